#!/bin/bash
# Round 3, GPU call 7: offset conv v3, LayerNorm(DYReLU) fused, 4-group dyrelu_coef, partial last tiles on the VLFuse image side --
# micro timings, device parity of the touched kernels, end-to-end A/Bs, default bench.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MQ_MICRO_ONLY=dyconv timeout 300 python tools/microbench.py gpurun_out/r03c7_micro_dyconv.json > gpurun_out/r03c7_micro_dyconv.log 2>&1; grep kernel gpurun_out/r03c7_micro_dyconv.log | cut -c1-200
timeout 600 python -m pytest tests -q -m gpu -k "test_block or opt_in or alternate or vlfuse" > gpurun_out/r03c7_pytest.log 2>&1; tail -6 gpurun_out/r03c7_pytest.log | cut -c1-400
for v in NONE=0 MQ_OFFSET_CONV_VARIANT=2 MQ_DYRELU_IN_LN=0 NONE=1; do
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r03c7_ab_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r03c7_ab_$v.log | cut -c1-140)"
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-experimental > gpurun_out/r03c7_bench_default.log 2>&1; tail -1 gpurun_out/r03c7_bench_default.log | cut -c1-200
