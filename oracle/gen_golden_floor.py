"""Generate tests/golden/floor_bench.json: the OPERAND FLOOR of the full-depth benchmark cases (test infrastructure).

    python -m oracle.gen_golden_floor            # ~10 minutes on 8 cores; deterministic (seeded weights / inputs)

For every benchmark case of tests/test_gpu_parity.py (full-depth MQ-GLIP-T on 800x1333 with the 81- and 141-token captions in fp16,
full-depth MQ-GLIP-L in bf16 and fp16) the fp32 oracle runs twice on the same seeded inputs: as is, and with ONLY the operands of its
contractions rounded to the 16-bit type (oracle/precision.py RoundGemmOperands -- accumulation, residual streams, normalisations,
softmaxes and GEMM outputs stay fp32).  The difference between the two is the smallest error ANY MFMA implementation with 16-bit
operands can have against the fp32 reference on these weights.  Stored per stage: max |err|, mean |err|, max normalised by
max(1, max |ref|), element count -- plus the fraction of the oracle's top detections the floor itself reproduces.
`parity_checks.check_benchmark_config` gates the product on these numbers (FLOOR_RATIO_MEDIAN / FLOOR_RATIO_ROW there):
the product's error may exceed the floor's by a stated factor and no more.  No reference code is involved (the oracle is pinned to
the reference elsewhere: tests/test_oracle_golden.py); the script that made the file is this one.
"""
import json
import os
import sys
import time
from dataclasses import replace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [("t", "short", ((800, 1333), (736, 1280)), torch.float16),
         ("t", "long", ((800, 1333),), torch.float16),
         ("l", "long", ((800, 1333),), torch.bfloat16),
         ("l", "long", ((800, 1333),), torch.float16)]


def main(out_path=None, only=None):
    import parity_checks as pc
    from oracle import detector as od, postprocess as opp
    from oracle.precision import RoundGemmOperands
    from oracle.weights import make_state_dict
    torch.set_num_threads(os.cpu_count())
    out_path = out_path or os.path.join(ROOT, "tests", "golden", "floor_bench.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    sds = {}
    for family, caption, hw, dtype in CASES:
        pc.use_dtype(dtype)                                            # inputs are rounded to the type under test, like the check does
        key = pc.bench_case_key(family, caption, hw, dtype)
        if only and only not in key:
            continue
        t0 = time.time()
        spec = pc.bench_spec(family)
        if family not in sds:
            sds[family] = make_state_dict(spec, 0)
        sd = sds[family]
        images, sizes, ids, am, pm, nv, bank = pc.bench_inputs(spec, caption, hw)
        with torch.no_grad():
            _, ref = od.forward(sd, spec, images, sizes, ids, am, pm, bank, return_intermediates=True)
            with RoundGemmOperands(dtype):
                _, fl = od.forward(sd, spec, images, sizes, ids, am, pm, bank, return_intermediates=True)
        rr, fr = pc.oracle_rows(ref, am, pm, nv), pc.oracle_rows(fl, am, pm, nv)
        rows = {}
        for name, (kind, r) in rr.items():
            f = fr[name][1]
            err = (f - r).abs()
            scale = max(1.0, r.abs().max().item())
            rows[name] = {"kind": kind, "max": err.max().item(), "mean": err.mean().item(), "norm": err.max().item() / scale,
                          "ref_absmax": r.abs().max().item(), "n": r.numel()}
        h, fh = ref["head"], fl["head"]
        for mode, (mdetr, ndet) in (("mdetr3000", (spec.mdetr_class_num, spec.detections_per_img)), ("dyhead", (-1, 100))):
            spec2 = replace(spec, mdetr_class_num=mdetr, detections_per_img=ndet)
            with torch.no_grad():
                od_ = opp.atss_postprocess(h["bbox_reg"], h["centerness"], h["dot_product_logits"], ref["anchors"], sizes, pm, spec2)
                fd_ = opp.atss_postprocess(fh["bbox_reg"], fh["centerness"], fh["dot_product_logits"], ref["anchors"], sizes, pm, spec2)
            top = 100 if ndet >= 300 else 50
            for b in range(len(hw)):
                frac = pc._match_detections(fd_[b]["boxes"], fd_[b]["scores"], fd_[b]["labels"], od_[b]["boxes"], od_[b]["scores"],
                                            od_[b]["labels"], top=top)
                rows[f"{mode}: detections img{b}"] = {"kind": "det", "norm": 1 - frac, "top": top}
        out[key] = rows
        print(f"{key}: {len(rows)} rows in {time.time() - t0:.0f} s; e.g. dot lvl0 mean {rows['dot-product logits lvl0']['mean']:.3e}", flush=True)
        with open(out_path, "w") as fh_:
            json.dump(out, fh_, indent=0, sort_keys=True)
    print("wrote", out_path)


if __name__ == "__main__":
    main(only=sys.argv[1] if len(sys.argv) > 1 else None)
