#!/bin/bash
# Round 4, GPU call 6: fused post-processing (csrc/post2.hip): device parity + end-to-end A/B; VLFuse image-side XCD mapping at B = 4 (MQ-GLIP-L)
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "post_fused or post_golden or check_full_model or score_agg or b8_graph or boundary_returns or check_nms or hip_graph_replay or vlfuse_kernels" > gpurun_out/r04c6_pytest.log 2>&1; grep -E "passed|failed|Error|assert|Fatal" gpurun_out/r04c6_pytest.log | tail -8 | cut -c1-800
for v in "MQ_POST_FUSED=0" "MQ_POST_FUSED=1" "MQ_POST_FUSED=0" "MQ_POST_FUSED=1"; do
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r04c6_ab_$v.log 2>&1
  echo "$v: rc=$? $(tail -1 gpurun_out/r04c6_ab_$v.log | cut -c1-200)"
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r04c6_ab_$v.log") if l.startswith("{")][-1])
    k=d["kernels_ms_per_step"]; print({n:k[n] for n in k if n.startswith("post") or n.startswith("align")})
except Exception as e: print("no json", e)
PY
done
timeout 300 python bench.py --workload mq-glip-l --steps 10 --warmup 3 --no-extras > gpurun_out/r04c6_glipl.log 2>&1; echo "mq-glip-l: $(tail -1 gpurun_out/r04c6_glipl.log | cut -c1-220)"
python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r04c6_glipl.log") if l.startswith("{")][-1])
    for r in d["rooflines"][:6]: print(r["kernel"][:50], r["ms_per_step"], r["frac"])
except Exception as e: print("no json", e)
PY
