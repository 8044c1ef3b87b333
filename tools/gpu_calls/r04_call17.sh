#!/bin/bash
# Round 4, GPU call 17: hardware queue count sweep (the default of the package is 8)
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
for q in 8 6 10 12 5 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r04c17_q$q.log 2>&1
  echo "GPU_MAX_HW_QUEUES=$q: rc=$? $(tail -1 gpurun_out/r04c17_q$q.log | cut -c1-200)"
done
