#!/usr/bin/env python
"""Where the replayed step spends its time, stage by stage: every stage of the bench.py workload (BASELINE configs[1]: B = 8, 800x1333, 141-token
caption) is captured as ITS OWN HIP graph and replayed -- Swin + FPN, the image-independent language front, the image-dependent language rest
(pre-select + GCP / BERT layers: the chain nothing else overlaps), VLDyHead, post-processing -- next to the whole forward.  The difference
between the sum of the serial stages and the whole step is what the side streams hide.

    python tools/stage_times.py [out.json]            MQ_* kernel switches apply (A/B of a stage in isolation)"""
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mq_det_amd import ops  # noqa: E402
from mq_det_amd.modeling import pipeline  # noqa: E402
from mq_det_amd.structures import ImageList  # noqa: E402


def graph_time(fn, iters=20):
    """fn() captured into a CUDA/HIP graph on a side stream (its own forks become graph branches), replayed `iters` times -> ms per replay."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    del keep
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda:0")
    ops.load_library()
    cfg, model, chunks = bench.build_model(dev)
    P = model._plan
    Bn = int(os.environ.get("MQ_STAGE_BATCH", "8"))
    H, W = bench.IMG_HW
    Hp, Wp = -(-H // 32) * 32, -(-W // 32) * 32
    imgs = torch.zeros(Bn, 3, Hp, Wp)
    imgs[:, :, :H, :W] = torch.randn(Bn, 3, H, W, generator=torch.Generator().manual_seed(1000))
    imgs = imgs.to(dev)
    caption, pmap = chunks[0]
    ids, am, max_kv = model.tokenize([caption] * Bn, dev)
    labels = [k for k, v in pmap.items() if len(v)]
    pm_key = tuple((k, tuple(pmap[k])) for k in labels)
    dtype = P["backbone.body.patch_embed.proj.weight"].dtype
    Tl = model._live_len(ids.shape[1], max_kv)
    ids, am = ids[:, :Tl].contiguous(), am[:, :Tl].contiguous()
    vision, idx = model.query_selector.select_cached(pm_key, labels, pmap, Bn, Tl, dev, dtype)
    from mq_det_amd.modeling.query_selector import build_token_index
    tokidx, label_ids = build_token_index(pmap, labels, dev)
    im_wh = torch.tensor([[W, H]] * Bn, dtype=torch.float32, device=dev)
    ops.activate(model._kernels)
    with torch.no_grad():
        feats, pooled = model._backbone_stage(imgs)
        front = pipeline.language_front(P, cfg, ids, am, True, max_kv=max_kv)
        lang = pipeline.language_backbone(P, cfg, ids, am, vision, pooled, idx, front=front, max_kv=max_kv, side_ok=True)
        lang["max_kv"] = max_kv
        head = pipeline.vldyhead(P, cfg, feats, lang)
        sizes = tuple(tuple(f.shape[-2:]) for f in feats)
        anchors = pipeline.grid_anchors(P, sizes, cfg.MODEL.RPN.ANCHOR_STRIDE, dev)
        out = {"batch": Bn, "text_rows": Tl, "kernel_selection": {k: v for k, v in ops.KERNELS.items() if v != ops.KERNEL_DEFAULTS.get(k)}}
        out["swin_fpn_ms"] = graph_time(lambda: model._backbone_stage(imgs))
        out["language_front_ms (beside Swin in the step)"] = graph_time(lambda: pipeline.language_front(P, cfg, ids, am, True, max_kv=max_kv))
        out["language_rest_ms (exposed chain)"] = graph_time(lambda: pipeline.language_backbone(P, cfg, ids, am, vision, pooled, idx, front=front, max_kv=max_kv, side_ok=True))
        out["language_rest_no_side_streams_ms"] = graph_time(lambda: pipeline.language_backbone(P, cfg, ids, am, vision, pooled, idx, front=front, max_kv=max_kv))
        out["pre_select_ms"] = graph_time(lambda: pipeline.pre_select(P, "language_backbone.body.model.pre_select", vision, pooled, cfg.VISION_QUERY.VISION_SCALE, side_ok=True))
        out["vldyhead_ms"] = graph_time(lambda: pipeline.vldyhead(P, cfg, feats, lang))
        cfg.MODEL.DYHEAD.LEVEL_STREAMS = False              # the same head with the text chain and the level work serialised on ONE stream
        out["vldyhead_single_stream_ms"] = graph_time(lambda: pipeline.vldyhead(P, cfg, feats, lang))
        cfg.MODEL.DYHEAD.LEVEL_STREAMS = True
        out["postprocess_ms"] = graph_time(lambda: pipeline.postprocess(cfg, dict(head), anchors, im_wh, tokidx, label_ids))
        tail = (ids, am, vision, idx, tokidx, label_ids, im_wh, max_kv)
        out["full_program_ms"] = graph_time(lambda: model._full_program(imgs, *tail))
    out["serial_sum_ms"] = out["swin_fpn_ms"] + out["language_rest_ms (exposed chain)"] + out["vldyhead_ms"] + out["postprocess_ms"]
    out = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items()}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
