#!/bin/bash
# Round 5, GPU call 6: settle the defaults of the two fused text kernels end to end (3 alternations x 60 steps: both on / both off / GCP only),
# and the head stage with and without its side streams (tools/stage_times.py).
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
  for env in "MQ_NONE=0" "MQ_GCP_ATTN_FUSED=0 MQ_BERT_ATTN_QKV_FUSED=0" "MQ_BERT_ATTN_QKV_FUSED=0"; do
    echo -n "$env: "; env $env timeout 90 python bench.py --steps 60 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r05c6_switch_ab.txt 2>&1; cat gpurun_out/r05c6_switch_ab.txt
MQ_GCP_ATTN_FUSED=0 MQ_BERT_ATTN_QKV_FUSED=0 timeout 300 python tools/stage_times.py gpurun_out/r05c6_stage_times_unfused.json 2>&1 | grep -v amdgpu.ids | tail -18
