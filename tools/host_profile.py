#!/usr/bin/env python3
"""GPU-box diagnostic: where does the HOST spend its time in a replayed step of the benchmark workload?  cProfile over N steps of bench.py's step()
(HIP-graph replay, caches off) + the wall time per step next to the GPU time per replay (HIP events around the replay only)."""
import cProfile
import os
import pstats
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mq_det_amd import ops  # noqa: E402
from mq_det_amd.structures import ImageList  # noqa: E402

dev = torch.device("cuda:0")
ops.load_library()
cfg, model, chunks = bench.build_model(dev)
B, (H, W) = 8, bench.IMG_HW
imgs = torch.zeros(B, 3, 800, 1344)
imgs[:, :, :H, :W] = torch.randn(B, 3, H, W)
images = ImageList(imgs.to(dev), [(H, W)] * B)
cap, pm = chunks[0]
caps = [cap] * B
for _ in range(5):
    out = model(images, captions=caps, positive_map=pm)
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t0 = time.perf_counter()
for _ in range(N):
    out = model(images, captions=caps, positive_map=pm)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N
# GPU time of one replay alone
ent = [e for e in model._graphs.values() if e.get("stage") == 2][0]
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a.record()
for _ in range(20):
    ent["graph"].replay()
b.record()
torch.cuda.synchronize()
print(f"wall per step {wall * 1e3:.3f} ms; graph replay alone (back to back) {a.elapsed_time(b) / 20:.3f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    out = model(images, captions=caps, positive_map=pm)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
