#!/bin/bash
# Round 4, GPU call 20: the bf16 twins of mq_attn_text_fwd / mq_patch_embed_fwd as their own rows of the bf16 list
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_bf16_block and (attention_text or patch_embed)" > gpurun_out/r04c20_pytest.log 2>&1; tail -6 gpurun_out/r04c20_pytest.log | cut -c1-300
