#!/bin/bash
# Round 4, GPU call 16: stream priorities of the side streams (text chain high / level streams high) -- end-to-end A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
for v in "MQ_STREAM_PRIORITY=none:0" "MQ_STREAM_PRIORITY=text:-1" "MQ_STREAM_PRIORITY=text:-1,levels:-1" "MQ_STREAM_PRIORITY=none:0" "MQ_STREAM_PRIORITY=text:-1"; do
  n=$(echo $v | tr ',:' '__')
  env $v timeout 300 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r04c16_ab_$n.log 2>&1
  echo "$v: rc=$? $(tail -1 gpurun_out/r04c16_ab_$n.log | cut -c1-200)"
done
