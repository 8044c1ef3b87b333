#!/bin/bash
# Round 4, GPU call 21: the grouped offset conv (csrc/conv_small3.hip, MQ_OFFSET_CONV_VARIANT=3) -- device parity (isolated body + bf16
# twin), its micro-benchmark against the per-level kernel, and the headline step with / without it on the same box
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
L=gpurun_out/r04c21.log
( timeout 120 python tests/test_gpu_parity.py offset_conv_group_kernel 2>&1 | tail -3
  timeout 60 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_bf16_block and conv3x3" 2>&1 | tail -2
  MQ_MICRO_ONLY=offset_conv timeout 60 python tools/microbench.py gpurun_out/r04c21_microbench_offset_conv.json 2>&1 | tail -6
  for v in 3 2 3; do
    echo "MQ_OFFSET_CONV_VARIANT=$v"; MQ_OFFSET_CONV_VARIANT=$v timeout 90 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done ) > $L 2>&1
cat $L | cut -c1-400
