"""Run every HIP-vs-oracle parity check and print / dump the full table (no fail-fast).
    python tests/gpu_diag.py [out.json]
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


def main():
    dev = torch.device("cuda:0")
    print("device:", torch.cuda.get_device_name(0), "| torch", torch.__version__, flush=True)
    rows = []
    for group, fn in pc.all_checks(dev):
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
