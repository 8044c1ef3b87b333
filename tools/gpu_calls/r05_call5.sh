#!/bin/bash
# Round 5, GPU call 5: where is the critical path now?  tools/stage_times.py (every stage of the step as its own replayed graph) under the
# default selection and with the fused text kernels off; HIP-graph tests with the new side streams.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "hip_graph_replay or backbone_and_caption or boundary or (bf16_block and check_gcp_block)" > gpurun_out/r05c5_pytest.log 2>&1; tail -4 gpurun_out/r05c5_pytest.log | cut -c1-300
timeout 300 python tools/stage_times.py gpurun_out/r05c5_stage_times.json 2>&1 | grep -v amdgpu.ids | tail -16
MQ_GCP_ATTN_FUSED=0 MQ_BERT_ATTN_QKV_FUSED=0 timeout 300 python tools/stage_times.py gpurun_out/r05c5_stage_times_unfused.json 2>&1 | grep -v amdgpu.ids | tail -16
MQ_BERT_ATTN_QKV_FUSED=0 timeout 300 python tools/stage_times.py gpurun_out/r05c5_stage_times_gcp_only.json 2>&1 | grep -v amdgpu.ids | tail -16
