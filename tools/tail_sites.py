#!/usr/bin/env python3
"""GPU-box diagnostic for the torch-native tail: ONE eager forward of the benchmark workload (caches off, no HIP graph) under a TorchDispatchMode;
every aten operator that is not a view is attributed to the innermost frame inside mq_det_amd/ (ops.py wrappers included) and printed with its
count and the bytes it touches.  Casts, copies and small elementwise passes are what `at::native::*` / `__amd_rocclr_copyBuffer` are in
profiles/r06_final_bench_kernel_stats.csv.      python tools/tail_sites.py [out.txt]"""
import collections
import os
import sys
import traceback

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mq_det_amd import ops  # noqa: E402
from mq_det_amd.structures import ImageList  # noqa: E402

VIEW = ("view", "reshape", "permute", "transpose", "slice", "select", "expand", "unsqueeze", "squeeze", "t.default", "alias", "as_strided", "detach",
        "unbind", "split", "_unsafe_view", "narrow", "chunk", "unflatten", "flatten", "empty", "lift_fresh", "_local_scalar_dense", "is_same_size",
        "sym_", "stride", "size", "record_stream", "is_pinned", "_has_compatible", "resize_")
LIB = ("mm.default", "addmm", "bmm", "linear", "matmul", "convolution")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()
        self.bytes = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        if any(v in name for v in VIEW) or any(v in name for v in LIB):
            return out
        site = None
        for fr in traceback.extract_stack()[:-1]:
            if "/mq_det_amd/" in fr.filename:
                site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
        if site is None:
            return out
        nb = sum(t.numel() * t.element_size() for t in ([out] if isinstance(out, torch.Tensor) else list(out) if isinstance(out, (tuple, list)) else [])
                 if isinstance(t, torch.Tensor))
        self.sites[(site, name)] += 1
        self.bytes[(site, name)] += nb
        return out


def main():
    dev = torch.device("cuda:0")
    ops.load_library()
    cfg, model, chunks = bench.build_model(dev)
    model.use_hip_graph = False                              # eager launches: the dispatch mode sees every operator
    B, (H, W) = 8, bench.IMG_HW
    imgs = torch.zeros(B, 3, 800, 1344)
    imgs[:, :, :H, :W] = torch.randn(B, 3, H, W)
    images = ImageList(imgs.to(dev), [(H, W)] * B)
    cap, pm = chunks[0]
    for _ in range(2):
        model(images, captions=[cap] * B, positive_map=pm)
    torch.cuda.synchronize()
    c = Census()
    with c, torch.no_grad():
        model(images, captions=[cap] * B, positive_map=pm)
    torch.cuda.synchronize()
    lines = [f"{sum(c.sites.values())} non-view, non-GEMM aten operators in one eager forward (B = {B})"]
    byop = collections.Counter()
    for (site, op), n in c.sites.items():
        byop[op] += n
    lines.append("by operator: " + ", ".join(f"{k} {v}" for k, v in byop.most_common(14)))
    for (site, op), n in sorted(c.sites.items(), key=lambda kv: -kv[1])[:70]:
        lines.append(f"{n:5d}  {c.bytes[(site, op)] / max(n, 1) / 1e6:8.2f} MB  {site:46s} {op}")
    print("\n".join(lines))
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
