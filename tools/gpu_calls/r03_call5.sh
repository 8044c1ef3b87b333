#!/bin/bash
# Round 3, GPU call 5: Swin MLP pass / tail split and the pair-split VLFuse image-side kernel (first device runs), the whole GPU
# suite under the median floor gate, default bench.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 400 python - <<PY > gpurun_out/r03c5_checks.log 2>&1
import sys, torch
sys.path.insert(0, "tests")
import parity_checks as pc
from mq_det_amd import ops
ops.load_library()
for name in ("check_swin_mlp", "check_vlfuse_kernels"):
    res = getattr(pc, name)(torch.device("cuda:0"))
    for r in res:
        print(("PASS " if r["ok"] else "FAIL ") + r["name"], "%.2e" % r["norm_err"])
    print(name, "ALL_OK" if all(r["ok"] for r in res) else "SOME_FAILED")
PY
grep "ALL_OK\|SOME_FAILED\|FAIL\|Error\|error" gpurun_out/r03c5_checks.log | head -12
MQ_MICRO_ONLY=swin timeout 300 python tools/microbench.py gpurun_out/r03c5_micro_swin.json > gpurun_out/r03c5_micro_swin.log 2>&1; grep kernel gpurun_out/r03c5_micro_swin.log | cut -c1-150
MQ_MICRO_ONLY=vlfuse timeout 300 python tools/microbench.py gpurun_out/r03c5_micro_vlfuse.json > gpurun_out/r03c5_micro_vlfuse.log 2>&1; grep kernel gpurun_out/r03c5_micro_vlfuse.log | cut -c1-200
for v in MQ_VLFUSE_I2T_VARIANT=1 MQ_SWIN_MLP2_FLAGS=1 NONE=0; do
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r03c5_ab_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r03c5_ab_$v.log | cut -c1-140)"
done
MQ_LADDER_OUT=$R/gpurun_out/r03c5_ladder.jsonl timeout 1500 python -m pytest tests -q -m gpu --durations=6 > gpurun_out/r03c5_pytest.log 2>&1; tail -25 gpurun_out/r03c5_pytest.log | cut -c1-500
( time timeout 900 python bench.py ) > gpurun_out/r03c5_bench_default.log 2> gpurun_out/r03c5_bench_default.time; tail -1 gpurun_out/r03c5_bench_default.log | cut -c1-300; tail -4 gpurun_out/r03c5_bench_default.time
