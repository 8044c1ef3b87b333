#!/bin/bash
# Round 6, GPU call 29: evidence at HEAD with KERNELS["BERT_CLAMP_FUSED"] = 1 as the default: the GPU suite as the driver runs it, smoke(), the default bench,
# per-step kernel statistics.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
( time MQ_LADDER_OUT=$R/gpurun_out/r06_final_gpu_suite_ladder.jsonl timeout 2400 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r06_final_gpu_suite.txt 2>&1; tail -6 gpurun_out/r06_final_gpu_suite.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_final_smoke.log 2>&1; tail -2 gpurun_out/r06_final_smoke.log | cut -c1-200
cd /tmp
for n in 4 24; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o bench -- python $R/bench.py --steps $n --warmup 2 --no-graph --no-extras > /dev/null 2>&1
  f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r06c29_kernel_stats_steps$n.csv
done
cd $R
python tools/per_step_kernel_stats.py gpurun_out/r06c29_kernel_stats_steps4.csv 4 gpurun_out/r06c29_kernel_stats_steps24.csv 24 gpurun_out/r06c29_per_step_kernel_stats.csv | tee gpurun_out/r06c29_per_step_summary.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_final_bench_default.json 2> gpurun_out/r06_final_bench_default.time; tail -1 gpurun_out/r06_final_bench_default.json | wc -c; tail -1 gpurun_out/r06_final_bench_default.json | cut -c1-700; tail -3 gpurun_out/r06_final_bench_default.time
cp bench_extras.json gpurun_out/r06_final_bench_default_extras.json 2>/dev/null
