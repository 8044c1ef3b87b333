#!/bin/bash
# Round 4, GPU call 4d: which capture topology crashes hipStreamEndCapture; do two graph replays on two streams overlap?
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
for v in "PROBE=prefork_joinall" "PROBE=nested_joinall" "PROBE=prefork_joinall PROBE_LANES=1 PROBE_SIDE=2" "PROBE=nested_joinall PROBE_LANES=1 PROBE_SIDE=1"; do
  n=$(echo $v | tr ' ' '_')
  env $v timeout 120 python tools/graph_probe.py > gpurun_out/r04c4_$n.log 2>&1
  echo "$v: rc=$? $(grep -E 'PROBE_OK|Error|error' gpurun_out/r04c4_$n.log | tail -4 | cut -c1-200)"
done
