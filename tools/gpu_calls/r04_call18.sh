#!/bin/bash
# Round 4, GPU call 18: number of side streams of the per-level DyConv work (4 = one per level P4..P7)
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
for n in 4 2 1 3 4; do
  MQ_LEVEL_SIDE_STREAMS=$n timeout 300 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r04c18_n$n.log 2>&1
  echo "MQ_LEVEL_SIDE_STREAMS=$n: rc=$? $(tail -1 gpurun_out/r04c18_n$n.log | cut -c1-200)"
done
