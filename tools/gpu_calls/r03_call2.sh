#!/bin/bash
# Round 3, GPU call 2: first device run of mq_swin_mlp2_fwd (4 flag variants) and mq_align_fused_fwd; the whole GPU suite with the promoted
# kernel set, the floor-ratio gate and the full-depth MQ-GLIP-L cases (every row of every check -> ladder jsonl); microbench of the new
# kernels against the ones they replace; default bench + A/Bs; kernel-trace stats of the default step.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
# 1. the new kernels first, each in a process of its own (a memory fault must not take the rest of the call down)
for k in check_swin_mlp check_align_fused; do
  timeout 300 python - <<PY > gpurun_out/r03c2_$k.log 2>&1
import sys, torch
sys.path.insert(0, "tests")
import parity_checks as pc
from mq_det_amd import ops
ops.load_library()
res = getattr(pc, "$k")(torch.device("cuda:0"))
for r in res:
    print(("PASS " if r["ok"] else "FAIL ") + r["name"], "%.2e" % r["norm_err"])
print("ALL_OK" if all(r["ok"] for r in res) else "SOME_FAILED")
PY
  echo "$k: $(tail -1 gpurun_out/r03c2_$k.log) ($(grep -c FAIL gpurun_out/r03c2_$k.log) failed)"; grep FAIL gpurun_out/r03c2_$k.log | head -5
done
# 2. microbench: Swin MLP v1 vs v2 x flags, align_fused vs the round-2 path
MQ_MICRO_ONLY=swin timeout 300 python tools/microbench.py gpurun_out/r03c2_micro_swin.json > gpurun_out/r03c2_micro_swin.log 2>&1; cat gpurun_out/r03c2_micro_swin.log | cut -c1-220
MQ_MICRO_ONLY=align timeout 300 python tools/microbench.py gpurun_out/r03c2_micro_align.json > gpurun_out/r03c2_micro_align.log 2>&1; cat gpurun_out/r03c2_micro_align.log | cut -c1-220
# 3. the GPU suite (every row -> ladder)
MQ_LADDER_OUT=$R/gpurun_out/r03c2_ladder.jsonl timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 > gpurun_out/r03c2_pytest.log 2>&1; tail -25 gpurun_out/r03c2_pytest.log | cut -c1-400
# 4. bench: default (= promoted set + new kernels), then single switches back
timeout 400 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-experimental > gpurun_out/r03c2_bench_default.log 2>&1; echo "default: $(tail -1 gpurun_out/r03c2_bench_default.log | cut -c1-140)"
for v in MQ_SWIN_MLP_VARIANT=1 MQ_SWIN_MLP2_FLAGS=0 MQ_SWIN_MLP2_FLAGS=3 MQ_ALIGN_FUSED=0; do
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r03c2_ab_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r03c2_ab_$v.log | cut -c1-140)"
done
# 5. kernel-trace stats of the default step, eager launches
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-graph --no-extras > $R/gpurun_out/r03c2_prof.log 2>&1
f=$(find /tmp/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r03c2_kernel_stats.csv && head -30 $f | cut -c1-170
