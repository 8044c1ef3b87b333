#!/bin/bash
# GPU call 13: validation of the load-hoisting rewrites (Swin MLP prologue / epilogue, window attention q/k/v + bias rows):
# whole GPU suite, default bench line, MQ-GroundingDINO bench.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 330 python -m pytest tests -q -m gpu > gpurun_out/r02_pytest13.log 2>&1; tail -4 gpurun_out/r02_pytest13.log | cut -c1-300
timeout 100 python bench.py --steps 10 --warmup 3 --no-lang-b64 --no-cpu-baseline > gpurun_out/r02_bench13_default.log 2>&1; tail -1 gpurun_out/r02_bench13_default.log | cut -c1-200
timeout 100 python bench.py --workload mq-gdino-t --steps 10 --warmup 3 > gpurun_out/r02_bench13_gdino_b16.log 2>&1; tail -1 gpurun_out/r02_bench13_gdino_b16.log | cut -c1-200
