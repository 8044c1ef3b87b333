#!/usr/bin/env python3
"""Run the GPU parity checks (tests/parity_checks.py, tests/gdino_checks.py) on CPU tensors through the kernel-source emulation
(tests/simt/): the development loop when no GPU minute is available -- edit a .hip file, run its check here, look at the ISA
(tools/isa_wait_scan.py, -Rpass-analysis=kernel-resource-usage), and only then go to the device.

    python tools/simt_checks.py [--bf16] [--schedule descending|random:<seed>] [--guard end|start] [--full] [--list] [group ...]

groups: the tags of parity_checks.all_checks (attention, window_attn, swin_fpn, gcp, bert, vlfuse, dcn, post, swin, gdino, roi, conv,
layernorm, dyconv, nms, swin-L, full, full-L, full-novq); no group = all.  Opt-in kernels are selected by their environment switches
(MQ_ATTN_RESIDENT=1, MQ_LN_VARIANT=2, MQ_OFFSET_CONV_VARIANT=2, MQ_PATCH_MERGE_FUSED=1).  Test infrastructure: the product never loads
the emulation library."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("groups", nargs="*")
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--schedule", default="ascending")
    ap.add_argument("--guard", choices=["end", "start"], default=None, help="every library argument against a guard page (SIGSEGV on an overrun)")
    ap.add_argument("--full", action="store_true", help="all cases of the sweeps (default: the quick subsets)")
    ap.add_argument("--list", action="store_true")
    args = ap.parse_args()
    import contextlib
    import torch
    import simt
    from simt import guard
    import parity_checks as pc
    from mq_det_amd.modeling import detector, gdino, pipeline, gdino_pipeline as gp
    cpu = torch.device("cpu")

    def prepare(self, device=None):
        self._validate_config()
        self._plan = pipeline.build_plan(self.state_dict(), self.cfg, cpu, dtype=detector.compute_dtype(self.cfg))
        self._plan_key, self.use_hip_graph = cpu, False
        return self._plan

    def prepare_gdino(self, device=None):
        self._plan = gp.build_gdino_plan(self.state_dict(), self.cfg, cpu, self._swin, dtype=detector.compute_dtype(self.cfg))
        self._plan_key, self.use_hip_graph = cpu, False
        return self._plan
    detector.GeneralizedVLRCNN_New.prepare, gdino.GroundingDINO.prepare = prepare, prepare_gdino
    pc.QUICK, pc.PINS = not args.full, False
    if args.bf16:
        pc.use_dtype(torch.bfloat16)
    checks = pc.all_checks(cpu)
    if args.list:
        print(" ".join(sorted({g for g, _ in checks})))
        return 0
    mode, _, seed = args.schedule.partition(":")
    n_bad = 0
    with simt.installed(), (guard.pointer_guard(args.guard) if args.guard else contextlib.nullcontext()), \
            (guard.guarded_ops(args.guard) if args.guard else contextlib.nullcontext()):
        simt.set_schedule(mode, int(seed or 0))
        for group, fn in checks:
            if args.groups and group not in args.groups:
                continue
            t = time.time()
            try:
                res = fn()
            except Exception as e:  # noqa: BLE001
                print(f"ERROR [{group}] {type(e).__name__}: {str(e)[:300]}", flush=True)
                n_bad += 1
                continue
            for r in (res if isinstance(res, list) else [res]):
                if "HIP-graph" in r["name"]:
                    continue                      # rows that assert an actual graph capture
                n_bad += not r["ok"]
                ev = r.get("elem_viol_frac", 0.0)
                print(f"{'PASS' if r['ok'] else 'FAIL'} [{group}] {r['name']} norm_err={r['norm_err']:.2e} tol={r['tol']:.1e} elem_viol={ev:.1e}{'' if r.get('elem_ok', True) else ' (ELEM)'}",
                      flush=True)
            print(f"   ... {time.time() - t:.1f} s", flush=True)
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
