#!/bin/bash
# Round 4, GPU call 11: rocprof kernel stats of the step at HEAD (eager, 5 + 2 + 3 steps), GLIP-L with the new key split
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-graph --no-extras > $R/gpurun_out/r04c11_prof.log 2>&1
f=$(find /tmp/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r04c11_kernel_stats.csv && head -5 $f | cut -c1-150
cd $R
timeout 300 python bench.py --workload mq-glip-l --steps 10 --warmup 3 --no-extras > gpurun_out/r04c11_glipl.log 2>&1; echo "mq-glip-l: $(tail -1 gpurun_out/r04c11_glipl.log | cut -c1-220)"
timeout 300 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r04c11_bench.log 2>&1; echo "default: $(tail -1 gpurun_out/r04c11_bench.log | cut -c1-220)"
