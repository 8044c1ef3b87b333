#!/usr/bin/env python
"""Stand-alone timings (HIP events, 20 launches after 3 warm-ups) of kernels that the configs[1] bench does not exercise or
that are worth seeing in isolation:  python tools/microbench.py [out.json]
  * mq_msdeform_attn_fwd at the MQ-GroundingDINO encoder shape (BASELINE configs[4]: B = 16, 4 levels of 800x1344, Q = 22 323)
  * mq_swin_mlp_fwd per Swin stage at B = 8
  * mq_window_attn_fwd with 144-token windows (Swin-L stage 1 at B = 4)
Rates are ALGORITHMIC bytes / flops over the event-measured launch time."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mq_det_amd import ops  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    dev = torch.device("cuda:0")
    ops.load_library()
    g = torch.Generator().manual_seed(0)
    out = []
    only = os.environ.get("MQ_MICRO_ONLY", "")
    if only in ("", "msda"):
        msda(dev, g, out)
    if only in ("", "swin"):
        swin(dev, g, out)
    if only in ("", "window"):
        window(dev, g, out)
    for r in out:
        print(json.dumps(r))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


def msda(dev, g, out):
    # ---- MSDeformAttn, encoder self-attention shape
    shapes = [(100, 168), (50, 84), (25, 42), (13, 21)]
    S = sum(h * w for h, w in shapes)
    B = 16
    v = torch.randn(B, S, 8, 32, generator=g).half().to(dev)
    ref = torch.rand(B, S, 1, 1, 1, 2, generator=g)
    loc = (ref + torch.randn(B, S, 8, 4, 4, 2, generator=g) * 0.02).clamp(0, 1).to(dev).contiguous()     # offsets of a few pixels
    attn = torch.rand(B, S, 8, 16, generator=g).softmax(-1).reshape(B, S, 8, 4, 4).to(dev).contiguous()
    ms = timeit(lambda: ops.ms_deform_attn(v, shapes, loc, attn))
    nb = v.numel() * 2 + loc.numel() * 4 + attn.numel() * 4 + B * S * 256 * 2
    gathered = B * S * 8 * 16 * 4 * 64                     # bytes the gather touches (4 corners x 64 B per sample), mostly L2 hits
    out.append({"kernel": "msda_kernel (encoder, B=16, Q=S=22323, 8 heads x 32, 4 levels x 4 points, fp16 values)", "ms": round(ms, 3),
                "algorithmic_GBs": round(nb / ms / 1e6, 1), "gather_GBs": round(gathered / ms / 1e6, 1),
                "algorithmic_bytes": nb, "frac_of_hbm_peak": round(nb / ms / 1e6 / 8000, 3)})


def swin(dev, g, out):
    # ---- fused Swin MLP per stage (B = 8, 800x1344)
    for C, M in ((96, 8 * 67200), (192, 8 * 16800), (384, 8 * 4200)):
        x = torch.randn(M, C, generator=g).to(dev)
        d = torch.randn(M, C, generator=g).half().to(dev)
        lg, lb = torch.ones(C).half().to(dev), torch.zeros(C).half().to(dev)
        w1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).half().to(dev)
        b1 = torch.zeros(4 * C).half().to(dev)
        w2p = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).half().to(dev)
        b2 = torch.zeros(C).half().to(dev)
        ms = timeit(lambda: ops.swin_mlp(x, d, lg, lb, 1e-5, w1, b1, w2p, b2, next_ln=(lg, lb, 1e-5)))
        fl, nb = 16.0 * M * C * C, M * C * (4 + 2 + 4 + 2)
        out.append({"kernel": f"swin_mlp_kernel C={C} M={M} variant={os.environ.get('MQ_SWIN_MLP_VARIANT', 'default')}", "ms": round(ms, 3), "TFLOPs": round(fl / ms / 1e9, 1),
                    "frac_of_mfma_peak": round(fl / ms / 1e9 / 2500, 3), "algorithmic_GBs": round(nb / ms / 1e6, 1),
                    "frac_of_hbm_peak": round(nb / ms / 1e6 / 8000, 3)})


def window(dev, g, out):
    # ---- window attention, 144-token windows (Swin-L stage 1, B = 4)
    Bn, H, W, C, heads, ws = 4, 200, 336, 192, 6, 12
    qkv = torch.randn(Bn, H, W, 3 * C, generator=g).half().to(dev)
    qb = torch.zeros(3 * C).half().to(dev)
    rel = ops.pad_rel_bias(torch.randn(heads, ws * ws, ws * ws, generator=g), ws).to(dev)
    for shift in (0, ws // 2):
        ms = timeit(lambda: ops.window_attention(qkv, qb, rel, heads, ws, shift))
        nb = qkv.numel() * 2 + Bn * H * W * C * 2
        out.append({"kernel": f"window_attn_kernel<10> ws=12 shift={shift} (Swin-L stage 1, B=4)", "ms": round(ms, 3),
                    "algorithmic_GBs": round(nb / ms / 1e6, 1), "frac_of_hbm_peak": round(nb / ms / 1e6 / 8000, 3)})


if __name__ == "__main__":
    main()
