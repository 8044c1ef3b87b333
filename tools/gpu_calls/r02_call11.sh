#!/bin/bash
# GPU call 11: whole GPU suite on the current tree (MSDeformAttn kernels v2, language front on a side stream, single-sync
# detection extraction), MQ-GroundingDINO bench, default bench line with the CPU baseline.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 420 python -m pytest tests -q -m gpu > gpurun_out/r02_pytest11.log 2>&1; tail -4 gpurun_out/r02_pytest11.log
timeout 120 python bench.py --workload mq-gdino-t --steps 10 --warmup 3 > gpurun_out/r02_bench11_gdino_b16.log 2>&1; tail -1 gpurun_out/r02_bench11_gdino_b16.log | cut -c1-200
timeout 100 python bench.py --workload mq-gdino-t --batch 1 --steps 10 --warmup 3 > gpurun_out/r02_bench11_gdino_b1.log 2>&1; tail -1 gpurun_out/r02_bench11_gdino_b1.log | cut -c1-200
timeout 240 python bench.py --steps 10 --warmup 3 > gpurun_out/r02_bench11_default.log 2>&1; tail -1 gpurun_out/r02_bench11_default.log | cut -c1-200
