#!/bin/bash
# Round 5, GPU call 3: (1) the precise-mode rows that failed in call 2 (patch_embed h1 slice in LDS, benchmark configuration); (2) the eight-wave
# version of mq_bert_attn_qkv_fwd: microbench with MQ_BERT_ATTN_OCC = 2 / 4, parity; (3) headline A/Bs: fused BERT attention on / off (both register
# budgets), the PLAIN instantiation of the DCNv2 kernel for the FPN convs on / off; (4) benchmark-configuration parity in fp16 with compaction.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MQ_LADDER_OUT=$R/gpurun_out/r05c3_f32_ladder.jsonl timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "f32 and (patch_embed or benchmark or bert_attn_qkv or full_model)" > gpurun_out/r05c3_pytest_f32.log 2>&1; tail -8 gpurun_out/r05c3_pytest_f32.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "check_bert_attn_qkv or check_swin_fpn or 141-token or check_dcn" > gpurun_out/r05c3_pytest_new.log 2>&1; tail -5 gpurun_out/r05c3_pytest_new.log | cut -c1-300
for occ in 2 4; do echo "MQ_BERT_ATTN_OCC=$occ"; MQ_BERT_ATTN_OCC=$occ MQ_MICRO_ONLY=bert_attn timeout 300 python tools/microbench.py gpurun_out/r05c3_micro_bert_attn_occ$occ.json 2>&1 | grep -v amdgpu.ids | cut -c1-330; done
for i in 1 2; do
  for env in "MQ_NONE=0" "MQ_BERT_ATTN_QKV_FUSED=0" "MQ_BERT_ATTN_OCC=4" "MQ_DCN_PLAIN=0"; do
    echo -n "$env: "; env $env timeout 90 python bench.py --steps 60 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r05c3_switch_ab.txt 2>&1; cat gpurun_out/r05c3_switch_ab.txt
