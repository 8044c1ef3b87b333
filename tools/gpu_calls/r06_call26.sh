#!/bin/bash
# Round 6, GPU call 26: Swin patch-merging reduction GEMM with fp32 output (no cast pass): parity of the backbone / full model in the three builds, benches.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "check_swin or check_full_model or benchmark_configuration_parity or glip_l" > gpurun_out/r06c26_pytest.log 2>&1; tail -3 gpurun_out/r06c26_pytest.log | cut -c1-300
for i in 1 2 3; do
  echo -n "fp16: "; timeout 120 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee gpurun_out/r06c26_bench.txt
