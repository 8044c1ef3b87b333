#!/bin/bash
# Round 4, GPU call 9: post-processing second generation (slices -> levels -> merge), patch embedding with coalesced stores: parity + timings
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "post_fused or post_golden or check_full_model or patch_embed or score_agg or b8_graph or boundary_returns or hip_graph_replay or swin_fpn" > gpurun_out/r04c9_pytest.log 2>&1; grep -E "passed|failed|Error|assert|Fatal" gpurun_out/r04c9_pytest.log | tail -8 | cut -c1-800
for v in "NONE=0" "NONE=1"; do
  env $v timeout 300 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r04c9_ab_$v.log 2>&1
  echo "$v: rc=$? $(tail -1 gpurun_out/r04c9_ab_$v.log | cut -c1-200)"
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r04c9_ab_$v.log") if l.startswith("{")][-1])
    k=d["kernels_ms_per_step"]; print({n:k[n] for n in k if n.startswith(("patch_embed","post","align"))})
except Exception as e: print("no json", e)
PY
done
timeout 300 python bench.py --workload mq-glip-l --steps 10 --warmup 3 --no-extras > gpurun_out/r04c9_glipl.log 2>&1; echo "mq-glip-l: $(tail -1 gpurun_out/r04c9_glipl.log | cut -c1-220)"
timeout 300 python bench.py --workload lvis --chunk-batch 32 --steps 2 --warmup 2 --no-extras > gpurun_out/r04c9_lvis.log 2>&1; echo "lvis: $(tail -1 gpurun_out/r04c9_lvis.log | cut -c1-220)"
