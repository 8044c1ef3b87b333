#!/bin/bash
# GPU call 7: first run of the MQ-GroundingDINO path: kernel features (attention byte mask, 4-head VLFuse + image mask, fused MSDeformAttn),
# shallow model vs oracle, then the full-depth 800x1333 model (error table only, no asserts yet).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export TMPDIR=/tmp
MQ_DIAG_ONLY=gdino timeout 900 python tests/gpu_diag.py gpurun_out/r02_gdino_diag7.json > gpurun_out/r02_gdino_diag7.log 2>&1
tail -60 gpurun_out/r02_gdino_diag7.log
MQ_DIAG_ONLY=gdino-bench timeout 900 python tests/gpu_diag.py gpurun_out/r02_gdino_bench_diag7.json > gpurun_out/r02_gdino_bench_diag7.log 2>&1
tail -30 gpurun_out/r02_gdino_bench_diag7.log
