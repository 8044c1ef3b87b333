#!/bin/bash
# Round 6, GPU call 27: same-box A/B of the Swin reduction GEMM with fp32 output (call 26 compared two different boxes); the strided-operand equality test.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "strided_operands" 2>&1 | tail -2
for i in 1 2 3; do
  for env in "MQ_SWIN_RED_F32OUT=0" "MQ_SWIN_RED_F32OUT=1"; do
    echo -n "fp16 $env: "; env $env timeout 120 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/r06c27_swin_red_ab.txt
