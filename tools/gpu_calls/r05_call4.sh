#!/bin/bash
# Round 5, GPU call 4: the two fused text kernels on the device.  (1) parity: mq_gcp_attn_fwd (fp16 / bf16 / fp32 operands), mq_bert_attn_qkv_fwd
# with prefetch distance two, the tiny full model and the GCP block through them; (2) microbenchmarks of both against the launches they replace;
# (3) headline A/Bs (60 steps, 2 alternations): default / MQ_GCP_ATTN_FUSED=0 / MQ_BERT_ATTN_QKV_FUSED=0; (4) the bench line with lang_path_b64.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MQ_LADDER_OUT=$R/gpurun_out/r05c4_ladder.jsonl timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "check_gcp_attn_fused or check_bert_attn_qkv or check_gcp_block or (test_block and check_full_model) or f32_full_model or bf16_block and check_full_model" > gpurun_out/r05c4_pytest.log 2>&1; tail -8 gpurun_out/r05c4_pytest.log | cut -c1-300
MQ_MICRO_ONLY=bert_attn timeout 300 python tools/microbench.py gpurun_out/r05c4_micro_bert_attn.json 2>&1 | grep -v amdgpu.ids | cut -c1-330
MQ_MICRO_ONLY=gcp_attn timeout 300 python tools/microbench.py gpurun_out/r05c4_micro_gcp_attn.json 2>&1 | grep -v amdgpu.ids | cut -c1-330
for i in 1 2; do
  for env in "MQ_NONE=0" "MQ_GCP_ATTN_FUSED=0" "MQ_BERT_ATTN_QKV_FUSED=0"; do
    echo -n "$env: "; env $env timeout 90 python bench.py --steps 60 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r05c4_switch_ab.txt 2>&1; cat gpurun_out/r05c4_switch_ab.txt
timeout 400 python bench.py --steps 20 --warmup 3 --no-experimental --no-cpu-baseline > gpurun_out/r05c4_bench.log 2>&1; tail -1 gpurun_out/r05c4_bench.log > gpurun_out/r05c4_bench.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05c4_bench.json'))
print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))
l=d.get('lang_path_b64',{})
print({k: l.get(k) for k in ('ms_language_path','language_path_frac_of_mfma_peak','attention_kernels_ms','attention_mfma_utilisation','bert_fused_launches','kernels_ms')})
PY
