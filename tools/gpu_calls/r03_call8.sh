#!/bin/bash
# Round 3, GPU call 8: VLFuse image side with Q in registers (129 .. 160 keys) and partially staged last tiles, ablation timings of the
# text-side kernel, end-to-end A/B of the image-side variants.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
MQ_MICRO_ONLY=vlfuse timeout 400 python tools/microbench.py gpurun_out/r03c8_micro_vlfuse.json > gpurun_out/r03c8_micro_vlfuse.log 2>&1; grep kernel gpurun_out/r03c8_micro_vlfuse.log | cut -c1-210
timeout 300 python -m pytest tests -q -m gpu -k "vlfuse or (test_block and vl)" > gpurun_out/r03c8_pytest.log 2>&1; tail -4 gpurun_out/r03c8_pytest.log | cut -c1-300
for v in MQ_VLFUSE_I2T_VARIANT=3 NONE=0 MQ_VLFUSE_I2T_VARIANT=3 NONE=1; do
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r03c8_ab_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r03c8_ab_$v.log | cut -c1-140)"
done
