#!/bin/bash
# Round 4, GPU call 7: mq_attn_text_fwd (BERT layers: one qkv GEMM + attention with V row-major): device parity, end-to-end A/B, B = 64 language path
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "attention_text or bert or check_full_model or benchmark_configuration_parity or hip_graph_replay" > gpurun_out/r04c7_pytest.log 2>&1; grep -E "passed|failed|Error|assert|Fatal" gpurun_out/r04c7_pytest.log | tail -8 | cut -c1-800
for v in "MQ_BERT_QKV_FUSED=0" "MQ_BERT_QKV_FUSED=1" "MQ_BERT_QKV_FUSED=0" "MQ_BERT_QKV_FUSED=1"; do
  env $v timeout 300 python bench.py --steps 30 --warmup 3 --no-experimental --no-cpu-baseline > gpurun_out/r04c7_ab_$v.log 2>&1
  echo "$v: rc=$? $(tail -1 gpurun_out/r04c7_ab_$v.log | cut -c1-200)"
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r04c7_ab_$v.log") if l.startswith("{")][-1])
    k=d["kernels_ms_per_step"]; print({n:k[n] for n in k if n.startswith("attn")})
    l=d["lang_path_b64"]; print({n:l[n] for n in ("ms_language_path","attention_kernels_ms","attention_mfma_utilisation","kernels_ms")})
except Exception as e: print("no json", e)
PY
done
