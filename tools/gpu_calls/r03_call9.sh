#!/bin/bash
# Round 3, GPU call 9: text side of VLFuse with the three-slot prefetch ring (microbench + device parity), default step twice.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
MQ_MICRO_ONLY=vlfuse timeout 300 python tools/microbench.py gpurun_out/r03c9_micro_vlfuse.json > gpurun_out/r03c9_micro_vlfuse.log 2>&1; grep kernel gpurun_out/r03c9_micro_vlfuse.log | grep -v ablation | cut -c1-210
timeout 300 python -m pytest tests -q -m gpu -k "vlfuse or (test_block and vl)" > gpurun_out/r03c9_pytest.log 2>&1; tail -3 gpurun_out/r03c9_pytest.log | cut -c1-300
for v in NONE=0 NONE=1; do
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r03c9_ab_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r03c9_ab_$v.log | cut -c1-140)"
done
