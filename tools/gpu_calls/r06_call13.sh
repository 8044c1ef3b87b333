#!/bin/bash
# Round 6, GPU call 13: DCNv2 LDS-DMA weights after the wait-count fixes (group 0: plain barriers, its gathers stay in flight; group 1: loop behind
# __restrict__ tile pointers -> no vmcnt(0) between the copy and the MFMA phase's LDS reads; s_waitcnt as a builtin hipcc's pass tracks; the copy as MUBUF buffer_load ... lds; MQ_DCN_BDMA_CNT=1: counted end-of-step wait).
# Parity on the device with the copy forced on (fp16 and split-precise), then A/B against register-staged weights, 3 alternations each.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
MQ_DCN_BDMA=1 MQ_DCN_BDMA_CNT=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "(check_dcn or check_dyconv or check_ref_pins or check_swin_fpn or check_full_model or fusion_layer or benchmark_configuration_parity) and not bf16 and not glip_l" > gpurun_out/r06c13_pytest_bdma.log 2>&1; tail -4 gpurun_out/r06c13_pytest_bdma.log | cut -c1-300
for i in 1 2 3; do
  for env in "MQ_DCN_BDMA=0" "MQ_DCN_BDMA=1" "MQ_DCN_BDMA=1 MQ_DCN_BDMA_CNT=1"; do
    echo -n "fp16 $env: "; env $env timeout 120 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  done
  for env in "MQ_DCN_BDMA=0" "MQ_DCN_BDMA=1" "MQ_DCN_BDMA=1 MQ_DCN_BDMA_CNT=1"; do
    echo -n "split $env: "; env $env timeout 200 python bench.py --dtype f32 --steps 8 --warmup 2 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done 2>&1 | tee gpurun_out/r06c13_bdma_ab.txt
