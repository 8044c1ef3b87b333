#!/usr/bin/env python3
"""GPU-box probe: does torch.mm(fp16, fp16, out_dtype=float32) exist here, and what does a 3-GEMM split linear cost against the fp32 library GEMM?"""
import time
import torch
dev = torch.device("cuda:0")
for (M, K, N) in ((33600, 384, 1152), (8400, 768, 2304), (1152, 768, 3072), (179200, 256, 256)):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    ref = (x.double() @ w.double().t())

    def split(t):
        hi = t.half()
        return hi, ((t - hi.float()) * 2048.0).half()
    whi, wlo = split(w)
    wcat = torch.cat([wlo, whi], 1).contiguous()

    def f32():
        return x @ w.t()

    def split3():
        xhi, xlo = split(x)
        xcat = torch.cat([xhi, xlo], 1)
        g1 = torch.mm(xhi, whi.t(), out_dtype=torch.float32)
        g2 = torch.mm(xcat, wcat.t(), out_dtype=torch.float32)
        return g1 + g2 * (1.0 / 2048.0)
    try:
        for name, fn in (("fp32 library GEMM", f32), ("3 x fp16 GEMM (split operands)", split3)):
            for _ in range(3):
                y = fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                y = fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            err = float((y.double() - ref).abs().max() / ref.abs().max())
            print(f"M={M} K={K} N={N}  {name:<34s} {dt * 1e3:8.3f} ms  {2 * M * K * N / dt / 1e12:7.1f} TFLOP/s(alg)  rel err {err:.2e}", flush=True)
    except Exception as e:  # noqa: BLE001
        print("split GEMM not available:", repr(e)[:300])
        break
