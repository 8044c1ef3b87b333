#!/bin/bash
# Round 5, GPU call 7: after switching the language side streams off again (calls 5 / 6: 21 ms steps): the headline and the fused-kernel A/B.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
  for env in "MQ_NONE=0" "MQ_GCP_ATTN_FUSED=0 MQ_BERT_ATTN_QKV_FUSED=0" "MQ_BERT_ATTN_QKV_FUSED=0" "MQ_LANG_SIDE_STREAMS=1"; do
    echo -n "$env: "; env $env timeout 90 python bench.py --steps 60 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r05c7_switch_ab.txt 2>&1; cat gpurun_out/r05c7_switch_ab.txt
