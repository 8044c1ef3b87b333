#!/bin/bash
# Round 6, evidence at HEAD (b): the GPU suite as the driver runs it (every row to a ladder file), smoke() as the driver runs it, and the default bench
# exactly as the driver runs it.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
( time MQ_LADDER_OUT=$R/gpurun_out/r06_final_gpu_suite_ladder.jsonl timeout 2400 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r06_final_gpu_suite.txt 2>&1; tail -6 gpurun_out/r06_final_gpu_suite.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_final_smoke.log 2>&1; tail -2 gpurun_out/r06_final_smoke.log | cut -c1-200
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_final_bench_default.json 2> gpurun_out/r06_final_bench_default.time; tail -1 gpurun_out/r06_final_bench_default.json | wc -c; tail -1 gpurun_out/r06_final_bench_default.json | cut -c1-2500; tail -3 gpurun_out/r06_final_bench_default.time
cp bench_extras.json gpurun_out/r06_final_bench_default_extras.json 2>/dev/null
