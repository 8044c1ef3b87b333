#!/usr/bin/env python3
"""DCNv2 grouped launch at the bench geometry (B = 8, five levels, 13 branches) with C = 128 / 256 / 384 / 512 input channels: the launch time
is  tiles x (fixed + k-steps x per-step); k-steps = 9 C / 64.  The fit separates the per-tile cost that does not shrink with the reduction
(sampling-state prologue, L2 warm-up, pipeline fill, epilogue, the tail of the last round of workgroups) from the k-loop.  GPU only.
    python tools/dcn_fixed_cost_probe.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mq_det_amd import ops  # noqa: E402


def timeit(fn, n=30, warm=5):
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
    g = torch.Generator().manual_seed(0)
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    B = 8
    rows = []
    for C in (128, 256, 384, 512):
        lv = [torch.randn(B, h, w, C, generator=g).half().to(dev) for h, w in sizes]
        om = [(torch.randn(B, 27, h, w, generator=g) * 1.5).to(dev) for h, w in sizes]
        wts = [ops.dcn_weight_tiles((torch.randn(256, 9 * C, generator=g) / 48).half().to(dev)) if ops.dcn_bdma() else
               (torch.randn(256, 9 * C, generator=g) / 48).half().to(dev) for _ in range(3)]
        bias = torch.zeros(256).half().to(dev)
        branches = []
        for l in range(5):
            spec = [(1, lv[l], 1)] + ([(2, lv[l - 1], 2)] if l > 0 else []) + ([(0, lv[l + 1], 1)] if l < 4 else [])
            for k, x, stride in spec:
                branches.append({"x": x, "om": om[l], "w": wts[k], "bias": bias, "stride": stride, "wy": None, "wx": None})
        ms = min(timeit(lambda: ops.dcnv2_group(branches, want_stats=True)) for _ in range(3))
        rows.append({"C": C, "ksteps": 9 * C // 64, "ms": round(ms, 4)})
        print(rows[-1], flush=True)
    # least squares  ms = a + b * ksteps
    n = len(rows)
    sx = sum(r["ksteps"] for r in rows); sy = sum(r["ms"] for r in rows)
    sxx = sum(r["ksteps"] ** 2 for r in rows); sxy = sum(r["ksteps"] * r["ms"] for r in rows)
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    a = (sy - b * sx) / n
    fit = {"fixed_ms": round(a, 4), "per_kstep_ms": round(b, 5), "fixed_share_at_C256": round(a / (a + 36 * b), 3), "bdma": bool(ops.dcn_bdma())}
    print(fit)
    if len(sys.argv) > 1:
        json.dump({"rows": rows, "fit": fit}, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
