#!/bin/bash
# Round 6, GPU call 11: are the library-level A/B switches of the fp16 step still set the right way on this stack?  (3 alternations, 40 steps each)
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
for i in 1 2 3; do
  for env in "MQ_NONE=0" "MQ_DCN_WAVES=8" "MQ_DCN_SYNC=2" "MQ_VLFUSE_QB=2" "MQ_BERT_CLAMP_FUSED=1" "MQ_SWIN_MLP_TAIL_STREAM=1"; do
    echo -n "$env: "; env $env timeout 120 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r06c11_switch_ab.txt 2>&1; cat gpurun_out/r06c11_switch_ab.txt
