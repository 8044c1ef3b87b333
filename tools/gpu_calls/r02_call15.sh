#!/bin/bash
# GPU call 15: MQ-GroundingDINO per-image feature cache (the replay checks now run through `_program_rest`) + extract_query refactor.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_gpu_parity.py -q -k "groundingdino_block or extract_query" > gpurun_out/r02_pytest15.log 2>&1; tail -3 gpurun_out/r02_pytest15.log | cut -c1-300
