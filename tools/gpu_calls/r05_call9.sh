#!/bin/bash
# Round 5, GPU call 9: the GPU suite exactly as the driver runs it (every row of every check to a ladder file), then the GCP microbenchmark with
# three weight-fragment register sets.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
( time MQ_LADDER_OUT=$R/gpurun_out/r05c9_ladder.jsonl timeout 1500 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r05c9_pytest.log 2>&1; tail -8 gpurun_out/r05c9_pytest.log | cut -c1-400
MQ_MICRO_ONLY=gcp_attn timeout 300 python tools/microbench.py gpurun_out/r05c9_micro_gcp_attn.json 2>&1 | grep -v amdgpu.ids | cut -c1-330
