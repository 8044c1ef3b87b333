#!/bin/bash
# Round 3, GPU call 10 (final state): DCNv2 ablation timings, the whole GPU suite, PMC traffic of the hot kernels (two passes),
# rocprofv3 kernel stats of the default step, default bench exactly as the driver runs it.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MQ_MICRO_ONLY=dcn timeout 200 python tools/microbench.py gpurun_out/r03c10_micro_dcn.json > gpurun_out/r03c10_micro_dcn.log 2>&1; grep kernel gpurun_out/r03c10_micro_dcn.log | cut -c1-200
MQ_LADDER_OUT=$R/gpurun_out/r03c10_ladder.jsonl timeout 1200 python -m pytest tests -q -m gpu --durations=5 > gpurun_out/r03c10_pytest.log 2>&1; tail -14 gpurun_out/r03c10_pytest.log | cut -c1-400
timeout 600 bash tools/pmc_traffic.sh > gpurun_out/r03c10_pmc.log 2>&1; ls -la gpurun_out/pmc_r03 | head
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-graph --no-extras > $R/gpurun_out/r03c10_prof.log 2>&1
f=$(find /tmp/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r03c10_kernel_stats.csv && head -12 $f | cut -c1-150
cd $R
( time timeout 900 python bench.py ) > gpurun_out/r03c10_bench_default.log 2> gpurun_out/r03c10_bench_default.time; tail -1 gpurun_out/r03c10_bench_default.log | cut -c1-300; tail -4 gpurun_out/r03c10_bench_default.time
