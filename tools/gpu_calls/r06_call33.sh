#!/bin/bash
# Round 6, GPU call 33: the layout guard of the pooled-token path (ops.pool2x2_tokens_supported) lets the benchmark pyramid through: launches of the kernel per forward.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python - <<'PY' 2>&1 | tail -3 | tee gpurun_out/r06c33_pooled_guard.txt
import torch, bench
from mq_det_amd import ops
from mq_det_amd.structures import ImageList
dev = torch.device("cuda:0")
ops.load_library()
cfg, model, chunks = bench.build_model(dev)
model.use_hip_graph = False
calls = {"n": 0}
real = ops.pool2x2_tokens
def counted(feats):
    calls["n"] += 1
    return real(feats)
ops.pool2x2_tokens = counted
B, (H, W) = 8, bench.IMG_HW
imgs = torch.zeros(B, 3, 800, 1344); imgs[:, :, :H, :W] = torch.randn(B, 3, H, W)
images = ImageList(imgs.to(dev), [(H, W)] * B)
cap, pm = chunks[0]
out = model(images, captions=[cap] * B, positive_map=pm)
torch.cuda.synchronize()
print("mq_pool2x2_tokens_fwd launches in one eager forward:", calls["n"], "| detections of image 0:", len(out[0].bbox))
PY
