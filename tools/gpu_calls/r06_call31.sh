#!/bin/bash
# Round 6, GPU call 31: the pooled FPN tokens of the GCP pre-select in one launch (mq_pool2x2_tokens_fwd, KERNELS["POOLED_TOKENS_FUSED"]): equality with the
# torch statement on the device, parity of the model with it on, same-box A/B.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "pooled_tokens or strided_operands" 2>&1 | tail -2
MQ_POOLED_TOKENS_FUSED=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "(check_pre_select or check_full_model or benchmark_configuration_parity or check_gcp) and not glip_l" > gpurun_out/r06c31_pytest.log 2>&1; tail -2 gpurun_out/r06c31_pytest.log | cut -c1-200
for i in 1 2 3; do
  for env in "MQ_POOLED_TOKENS_FUSED=0" "MQ_POOLED_TOKENS_FUSED=1"; do
    echo -n "fp16 $env: "; env $env timeout 120 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/r06c31_pooled_ab.txt
