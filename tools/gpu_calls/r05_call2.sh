#!/bin/bash
# Round 5, GPU call 2: (1) the precise-mode tests again after the fixes of call 1 (+ the new fused BERT attention in fp32); (2) the new kernel and
# the live-row compaction on the device: parity (fp16 / bf16), microbench against the round-4 path, the language path at B = 64;
# (3) headline A/B: default (compaction + fused BERT attention + grouped epilogue) / MQ_BERT_ATTN_QKV_FUSED=0 / compaction off.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MQ_LADDER_OUT=$R/gpurun_out/r05c2_f32_ladder.jsonl timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "f32" > gpurun_out/r05c2_pytest_f32.log 2>&1; tail -12 gpurun_out/r05c2_pytest_f32.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "check_bert_attn_qkv or test_bert_layer or check_full_model or check_fusion_layer or boundary or hip_graph_replay or backbone_and_caption" > gpurun_out/r05c2_pytest_new.log 2>&1; tail -6 gpurun_out/r05c2_pytest_new.log | cut -c1-300
MQ_MICRO_ONLY=bert_attn timeout 300 python tools/microbench.py gpurun_out/r05c2_micro_bert_attn.json 2>&1 | tail -6
for i in 1 2; do
  for env in "MQ_NONE=0" "MQ_BERT_ATTN_QKV_FUSED=0" "MQ_COMPACT_TEXT=0"; do
    echo -n "$env: "; env $env timeout 90 python bench.py --steps 60 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r05c2_switch_ab.txt 2>&1; cat gpurun_out/r05c2_switch_ab.txt
timeout 400 python bench.py --steps 20 --warmup 3 --no-experimental --no-cpu-baseline > gpurun_out/r05c2_bench.log 2>&1; tail -1 gpurun_out/r05c2_bench.log > gpurun_out/r05c2_bench.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05c2_bench.json'))
print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))
print(json.dumps(d.get('lang_path_b64'))[:1500])
PY
