#!/bin/bash
# Round 6, GPU call 16: DCNv2 with LDS-copied weights, 16 waves (wave tile 32 x 64, the default) against 8 waves (64 x 64: a third fewer fragment
# reads per k-step, half the waves to hide latency), fp16 operands, 3 alternations; parity of the 8-wave variant first.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
MQ_DCN_WAVES=8 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "(check_dcn or check_dyconv or check_swin_fpn or check_full_model) and not bf16 and not f32" > gpurun_out/r06c16_pytest_w8.log 2>&1; tail -3 gpurun_out/r06c16_pytest_w8.log | cut -c1-300
for i in 1 2 3; do
  for env in "MQ_DCN_WAVES=16" "MQ_DCN_WAVES=8"; do
    echo -n "fp16 $env: "; env $env timeout 120 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  done
done 2>&1 | tee gpurun_out/r06c16_dcn_waves_ab.txt
