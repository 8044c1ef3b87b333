#!/bin/bash
# Round 4, GPU call 15: the default bench exactly as the driver runs it (duration check after the CPU-baseline fix) + smoke()
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
( time timeout 900 python bench.py ) > gpurun_out/r04c15_bench_default.log 2> gpurun_out/r04c15_bench_default.time; tail -1 gpurun_out/r04c15_bench_default.log | cut -c1-200; tail -3 gpurun_out/r04c15_bench_default.time
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r04c15_smoke.log 2>&1; tail -4 gpurun_out/r04c15_smoke.log
