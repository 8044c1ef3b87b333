#!/bin/bash
# Round 6, GPU call 30: per-STEP kernel statistics of the eager fp16 step (difference of two rocprofv3 runs with 4 and 24 timed steps).
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for n in 4 24; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o bench -- python $R/bench.py --steps $n --warmup 2 --no-graph --no-extras > $R/gpurun_out/r06c30_prof_$n.log 2>&1
  f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r06c30_kernel_stats_steps$n.csv
done
cd $R
python tools/per_step_kernel_stats.py gpurun_out/r06c30_kernel_stats_steps4.csv 4 gpurun_out/r06c30_kernel_stats_steps24.csv 24 gpurun_out/r06c30_per_step_kernel_stats.csv | tee gpurun_out/r06c30_per_step_summary.txt
