#!/usr/bin/env python3
"""Count the torch-native operators the host glue issues per forward, by call site (TEST / DEVELOPMENT TOOL: runs the product's
modeling/pipeline.py on CPU with the HIP entry points replaced by tests/ops_emulation.py and a TorchDispatchMode that attributes every
aten call made OUTSIDE an emulated entry point to the innermost pipeline.py line).  On the GPU each of those is a launch of an
at::native kernel (or a hipBLASLt GEMM): the list is the map of the "torch-native tail" of DESIGN.md section 15 item 6.

    python tools/glue_ops.py [--stage language|head|all]
"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

VIEW_OPS = ("view", "reshape", "permute", "transpose", "slice", "select", "expand", "unsqueeze", "squeeze", "t.default", "alias", "as_strided",
            "detach", "unbind", "split", "_unsafe_view", "narrow", "chunk", "unflatten", "flatten", "empty", "lift_fresh", "_local_scalar_dense",
            "is_same_size", "sym_", "stride", "size")


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()
        self.ops = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(v in name for v in VIEW_OPS):
            return out
        site, inside_emulation = None, False
        for fr in traceback.extract_stack()[:-1]:
            if fr.filename.endswith("ops_emulation.py"):
                inside_emulation = site is not None
                if inside_emulation:
                    break
            if fr.filename.endswith(("modeling/pipeline.py", "modeling/detector.py")):
                site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
        if site is None or inside_emulation:
            return out
        self.sites[(site, name.replace("aten.", ""))] += 1
        self.ops[name.replace("aten.", "")] += 1
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="all")
    args = ap.parse_args()
    import ops_emulation as emu
    import parity_checks as pc
    from oracle import tiny_spec
    from oracle.weights import make_state_dict
    from mq_det_amd import get_cfg, ops as real_ops
    from mq_det_amd.modeling import pipeline
    from mq_det_amd.modeling.query_selector import QuerySelector
    spec = tiny_spec()
    sd = make_state_dict(spec, 0)
    cfg = get_cfg()
    cfg.MODEL.SWINT.DEPTHS = spec.swin_depths
    cfg.MODEL.LANGUAGE_BACKBONE.NUM_HIDDEN_LAYERS = spec.bert_layers
    cfg.MODEL.LANGUAGE_BACKBONE.QV_START = spec.qv_start
    cfg.MODEL.LANGUAGE_BACKBONE.BERT_VOCAB_SIZE = spec.vocab
    cfg.MODEL.DYHEAD.NUM_CONVS = spec.dyhead_convs
    cfg.MODEL.DYHEAD.NUM_CLASSES = spec.num_classes
    cfg.MODEL.ATSS.DETECTIONS_PER_IMG = spec.detections_per_img
    P = pipeline.build_plan(sd, cfg, torch.device("cpu"), dtype=torch.float16)
    pipeline.ops = emu.namespace(real_ops)
    images, sizes, ids, am, pm, bank = pc.make_inputs(spec)
    with torch.no_grad():
        x = images.half().contiguous(memory_format=torch.channels_last)
        feats = pipeline.fpn_forward(P, pipeline.swin_forward(P, cfg, x))
        qs = QuerySelector(cfg)
        qs.load_query_bank(bank)
        labels = [k for k, v in pm.items() if len(v)]
        vision, idx = qs.select([labels] * 2, [pm] * 2, ids.shape[1], torch.device("cpu"), torch.float16)
        pooled = pipeline.pooled_fpn_tokens(feats)
        front = pipeline.language_front(P, cfg, ids, am, True)
        stages = {}
        c = Counter()
        with c:
            lang = pipeline.language_backbone(P, cfg, ids, am, vision, pooled, idx, front=front)
        stages["language (image-dependent half)"] = c
        c = Counter()
        with c:
            pipeline.vldyhead(P, cfg, feats, lang)
        stages["head"] = c
    print(f"tiny model: {spec.bert_layers} BERT layers (vision queries from layer {spec.qv_start}), {spec.dyhead_convs} fusion layers")
    for name, c in stages.items():
        if args.stage not in ("all", name.split()[0]):
            continue
        print(f"\n== {name}: {sum(c.ops.values())} torch operators outside the HIP entry points")
        print("   by operator: " + ", ".join(f"{k} {v}" for k, v in c.ops.most_common(16)))
        for (site, op), n in sorted(c.sites.items(), key=lambda kv: (kv[0][0].split()[0].split(':')[0], int(kv[0][0].split()[0].split(':')[1]), kv[0][1])):
            print(f"   {n:4d}  {site:48s} {op}")


if __name__ == "__main__":
    main()
