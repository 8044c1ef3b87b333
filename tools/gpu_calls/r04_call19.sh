#!/bin/bash
# Round 4, GPU call 19: the GPU suite exactly as the driver runs it at round end (final HEAD)
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04c19_pytest.log 2>&1; tail -4 gpurun_out/r04c19_pytest.log | cut -c1-300
