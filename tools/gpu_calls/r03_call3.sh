#!/bin/bash
# Round 3, GPU call 3: deep-prefetch Swin MLP (first device run of the counted-vmcnt pipeline), the whole GPU suite without -x
# (floor gate, bf16 list, full-depth MQ-GLIP-L, 19-argument DCN stub, tie rule), flag A/Bs, and the default bench exactly as the driver runs it.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python - <<PY > gpurun_out/r03c3_check_swin_mlp.log 2>&1
import sys, torch
sys.path.insert(0, "tests")
import parity_checks as pc
from mq_det_amd import ops
ops.load_library()
res = pc.check_swin_mlp(torch.device("cuda:0"))
for r in res:
    print(("PASS " if r["ok"] else "FAIL ") + r["name"], "%.2e" % r["norm_err"])
print("ALL_OK" if all(r["ok"] for r in res) else "SOME_FAILED")
PY
echo "check_swin_mlp: $(tail -1 gpurun_out/r03c3_check_swin_mlp.log) ($(grep -c FAIL gpurun_out/r03c3_check_swin_mlp.log) failed)"; grep FAIL gpurun_out/r03c3_check_swin_mlp.log | head -5
MQ_MICRO_ONLY=swin timeout 300 python tools/microbench.py gpurun_out/r03c3_micro_swin.json > gpurun_out/r03c3_micro_swin.log 2>&1; grep kernel gpurun_out/r03c3_micro_swin.log | cut -c1-160
MQ_LADDER_OUT=$R/gpurun_out/r03c3_ladder.jsonl timeout 1500 python -m pytest tests -q -m gpu --durations=6 > gpurun_out/r03c3_pytest.log 2>&1; tail -30 gpurun_out/r03c3_pytest.log | cut -c1-600
for v in MQ_SWIN_MLP2_FLAGS=1 MQ_SWIN_MLP2_FLAGS=0 MQ_SWIN_MLP2_FLAGS=3 MQ_SWIN_MLP2_FLAGS=2; do
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r03c3_ab_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r03c3_ab_$v.log | cut -c1-140)"
done
( time timeout 900 python bench.py ) > gpurun_out/r03c3_bench_driver_style.log 2> gpurun_out/r03c3_bench_driver_style.time; tail -1 gpurun_out/r03c3_bench_driver_style.log | cut -c1-300; tail -4 gpurun_out/r03c3_bench_driver_style.time
