#!/bin/bash
# Round 4, GPU call 8: VLFuse softmax diet (both directions), mq_patch_embed_fwd (16-bit channels-last and fp32 NCHW pixels): device parity,
# micro-benchmarks, end-to-end A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "patch_embed or vlfuse_kernels or check_vl_fuse or fusion_layer or check_full_model or swin_fpn or boundary_returns or hip_graph_replay or b8_graph" > gpurun_out/r04c8_pytest.log 2>&1; grep -E "passed|failed|Error|assert|Fatal" gpurun_out/r04c8_pytest.log | tail -8 | cut -c1-800
MQ_MICRO_ONLY=vlfuse timeout 200 python tools/microbench.py gpurun_out/r04c8_micro_vlfuse.json > gpurun_out/r04c8_micro_vlfuse.log 2>&1; grep -E "default|Q tile" gpurun_out/r04c8_micro_vlfuse.log | cut -c1-220
for v in "MQ_PATCH_EMBED_FUSED=0" "MQ_PATCH_EMBED_FUSED=1" "MQ_PATCH_EMBED_FUSED=0" "MQ_PATCH_EMBED_FUSED=1"; do
  env $v timeout 300 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r04c8_ab_$v.log 2>&1
  echo "$v: rc=$? $(tail -1 gpurun_out/r04c8_ab_$v.log | cut -c1-200)"
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r04c8_ab_$v.log") if l.startswith("{")][-1])
    k=d["kernels_ms_per_step"]; print({n:k[n] for n in k if n.startswith(("patch","vlfuse","layernorm_c96"))})
except Exception as e: print("no json", e)
PY
done
