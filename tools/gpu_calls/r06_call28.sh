#!/bin/bash
# Round 6, GPU call 28: the fused clamp / GELU forms of the fusion-layer BERT copies (KERNELS["BERT_CLAMP_FUSED"]) re-checked on the final step
# (call 11: 0.6 % slower on round 5's kernels): same-box A/B, 3 alternations.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
  for env in "MQ_BERT_CLAMP_FUSED=0" "MQ_BERT_CLAMP_FUSED=1"; do
    echo -n "fp16 $env: "; env $env timeout 120 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/r06c28_clamp_fused_ab.txt
