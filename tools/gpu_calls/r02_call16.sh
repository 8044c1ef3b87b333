#!/bin/bash
# GPU call 16: the INTEGRATION.md operator stubs executed as written against the oracle.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
timeout 80 python -m pytest tests/test_gpu_parity.py -q -k "integration_md" > gpurun_out/r02_pytest16.log 2>&1; tail -15 gpurun_out/r02_pytest16.log | cut -c1-300
