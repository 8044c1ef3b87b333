#!/bin/bash
# Round 5, GPU call 8: weight streams with their loads pinned in flight (scheduling fences): the Swin MLP tail kernel (ring of 12, was ~2-3 in
# flight) and the GEMMs of mq_gcp_attn_fwd (a whole group, was ~2); parity of both; microbenchmarks; headline.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "(test_block and (check_swin_mlp or check_gcp_attn_fused or check_swin_fpn)) or (bf16_block and check_swin_mlp)" > gpurun_out/r05c8_pytest.log 2>&1; tail -3 gpurun_out/r05c8_pytest.log | cut -c1-300
MQ_MICRO_ONLY=swin timeout 300 python tools/microbench.py gpurun_out/r05c8_micro_swin.json 2>&1 | grep -v amdgpu.ids | grep "C=384" | cut -c1-200
MQ_MICRO_ONLY=gcp_attn timeout 300 python tools/microbench.py gpurun_out/r05c8_micro_gcp_attn.json 2>&1 | grep -v amdgpu.ids | cut -c1-330
for i in 1 2; do
  for env in "MQ_NONE=0" "MQ_GCP_ATTN_FUSED=0 MQ_BERT_ATTN_QKV_FUSED=0"; do
    echo -n "$env: "; env $env timeout 90 python bench.py --steps 60 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r05c8_switch_ab.txt 2>&1; cat gpurun_out/r05c8_switch_ab.txt
