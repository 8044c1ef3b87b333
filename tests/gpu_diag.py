"""Run every HIP-vs-oracle parity check and print / dump the full table (no fail-fast).
    python tests/gpu_diag.py [out.json]
    python tests/gpu_diag.py --ladder out.txt     per-stage error ladder of the full-depth model at the benchmark
                                                  configuration: product (fp32 and fp16 residual streams) beside the
                                                  fp16-operand floor of the oracle (oracle/precision.py)
"""
import json
import os
import sys
import time
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_checks as pc  # noqa: E402


def ladder(path):
    dev = torch.device("cuda:0")
    lines = [f"device: {torch.cuda.get_device_name(0)} | torch {torch.__version__}",
             "Full-depth MQ-GLIP-T (Swin 2-2-6-2, 12 BERT + 6 GCP, 6 fusion layers), one 800x1333 image, 141-token caption, vs the fp32 CPU oracle.",
             "norm = max|err| / max(1, max|ref|); mean = mean|err|.  floor = the oracle with ONLY the contraction operands rounded to fp16",
             "(smallest error any fp16-operand MFMA implementation can have on these weights).  1e-3 is the north-star tolerance.", ""]
    t = time.time()
    a = pc.check_benchmark_config(dev, "long", ((800, 1333),), residual_fp32=True, floor=True)
    b = pc.check_benchmark_config(dev, "long", ((800, 1333),), residual_fp32=False, floor=False)
    lines.append(f"{'stage':<58s} {'fp32 streams: norm':>18s} {'mean':>9s} | {'fp16 streams: norm':>18s} {'mean':>9s} | {'floor: norm':>11s} {'mean':>9s} | x floor")
    for ra, rb in zip(a, b):
        name = ra["name"].split("] ", 1)[1]
        if "floor_norm_err" in ra:
            ratio = ra["mean_err"] / max(ra["floor_mean_err"], 1e-12)
            lines.append(f"{name:<58s} {ra['norm_err']:18.2e} {ra['mean_err']:9.2e} | {rb['norm_err']:18.2e} {rb['mean_err']:9.2e} | "
                         f"{ra['floor_norm_err']:11.2e} {ra['floor_mean_err']:9.2e} | {ratio:5.2f}")
        else:
            lines.append(f"{name:<58s} {ra['norm_err']:18.2e} {'':>9s} | {rb['norm_err']:18.2e}")
    lines.append(f"\n({time.time() - t:.0f} s)")
    txt = "\n".join(lines)
    print(txt)
    with open(path, "w") as f:
        f.write(txt + "\n")
    return 0


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--ladder":
        return ladder(sys.argv[2])
    dev = torch.device("cuda:0")
    print("device:", torch.cuda.get_device_name(0), "| torch", torch.__version__, flush=True)
    rows = []
    only = os.environ.get("MQ_DIAG_ONLY")                       # e.g. MQ_DIAG_ONLY=gdino: one group of checks
    checks = [(g, f) for g, f in pc.all_checks(dev) if not only or g in only.split(",")]
    if only and "gdino-bench" in only.split(","):
        import gdino_checks as gc
        checks.append(("gdino-bench", lambda: gc.check_gdino_benchmark_config(dev)))
    for group, fn in checks:
        t = time.time()
        try:
            r = fn()
            torch.cuda.synchronize()
            r = r if isinstance(r, list) else [r]
        except Exception as e:  # noqa: BLE001
            traceback.print_exc()
            r = [{"name": f"{group}: EXCEPTION {type(e).__name__}: {str(e)[:200]}", "max_err": float("nan"),
                  "mean_err": float("nan"), "ref_absmax": 1.0, "norm_err": float("nan"), "tol": 0, "ok": False}]
        for x in r:
            x["group"] = group
            x["secs"] = round(time.time() - t, 2)
            rows.append(x)
            print(f"[{'PASS' if x['ok'] else 'FAIL'}] {x['name']:<75s} max_err={x['max_err']:.3e} mean={x['mean_err']:.2e} "
                  f"ref_absmax={x['ref_absmax']:.3g} norm={x['norm_err']:.2e} (tol {x['tol']:.0e})", flush=True)
    nfail = sum(not x["ok"] for x in rows)
    print(f"\n{len(rows) - nfail}/{len(rows)} checks pass")
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(rows, f, indent=1)
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
