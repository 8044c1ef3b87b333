#!/bin/bash
# GPU call 10: MSDeformAttn fused kernel v2 (branch-free gather: 16 corner loads in flight instead of 64 serialised round trips;
# 63 VGPRs = 8 waves / SIMD; XCD-contiguous query order): parity of the kernel + model, then bench A/B of the query order.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "groundingdino_block or msdeform" > gpurun_out/r02_pytest10.log 2>&1; tail -3 gpurun_out/r02_pytest10.log
timeout 120 python bench.py --workload mq-gdino-t --steps 10 --warmup 3 > gpurun_out/r02_bench10_gdino_b16.log 2>&1; tail -1 gpurun_out/r02_bench10_gdino_b16.log | cut -c1-200
MQ_MSDA_ORDER=0 timeout 120 python bench.py --workload mq-gdino-t --steps 10 --warmup 3 > gpurun_out/r02_bench10_gdino_b16_order0.log 2>&1; tail -1 gpurun_out/r02_bench10_gdino_b16_order0.log | cut -c1-200
