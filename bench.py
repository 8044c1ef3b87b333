#!/usr/bin/env python
"""bench.py -- images/sec of the MQ-GLIP-T vision-language forward on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json configs[1]: one "step" = one `model(images, captions, positive_map)` forward (Swin -> FPN ->
BERT+GCP -> VLDyHead -> ATSS post-processing to list[BoxList]) on a batch of 8 synthetic 800x1333 images per GPU, MQ-GLIP-T,
5 vision queries per class, a 40-class caption of 141 tokens (the length of an LVIS chunk caption), fp16 operands, inputs
resident in HBM, followed (N > 1) by the fixed-shape RCCL all-gather of the detections.  Weak scaling: per-GPU batch fixed.
EVERY step computes the whole forward: the per-image feature cache and the per-caption language cache of the boundary
 are switched OFF for this workload -- the synthetic loop re-sends the same tensors, and skipping the
backbone would be work skipped inside the timed region.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline       the hand-written kernel that takes the most time per step (timed live with HIP events on the launch stream)
  rooflines      the same record for every hot hand-written kernel: MFMA-bound (TFLOP/s vs 2.5 PFLOP/s dense fp16) with the
                 ALGORITHMIC and the EXECUTED flop counts side by side, HBM-bound (algorithmic GB/s vs 8 TB/s)
  lang_path_b64  north-star target line: the language path (12 BERT + 6 GCP + pre-select) at B = 64 on one GPU, with the
                 MFMA utilisation of its attention kernels
  kernel_set_ab  the same workload once more (bounded subprocess) on the ROUND-2 kernel set -- every operator with two
                 implementations on the one that is not the default (ops.KERNEL_DEFAULTS) -- i.e. what round 3's selection buys
  other_configs  the other BASELINE.json configurations as bounded subprocess lines of this same script: configs[2] shape
                 (--workload lvis, chunk batching), configs[3] (--workload mq-glip-l, bf16), configs[4] (--workload mq-gdino-t, B = 16)
  cpu_baseline   (N = 1) BASELINE.md section 3 / configs[0]: the fp32 CPU oracle, GLIP-T without vision queries, one 800x1333
                 image, 20-token caption, 2 warm-up + 5 timed forwards, median.

    --workload lvis   BASELINE.json configs[2] shape on this many GPUs: 1203 synthetic categories -> 31 chunk captions of
                      <= 40 classes, each step = a NEW batch of images x 31 forwards (the LVIS protocol of
                      engine/inference.py:605-625) with the boundary's caches ON; value = forwards/s, plus LVIS-style
                      images/s = forwards/s / 31.
"""
import argparse
import json
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")        # mq_det_amd.RECOMMENDED_ENV: read when the HIP runtime initialises, so before `import torch`

import statistics  # noqa: E402
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 8
IMG_HW = (800, 1333)
NUM_CLASSES_IN_CAPTION = 40
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
# algorithmic work of one image-forward (2*MAC), BASELINE.md section 2 / SURVEY.md 8(d)
GFLOP_PER_IMAGE = 1451.0
MFMA_PEAK_TFLOPS = 2500.0          # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


GFLOP_PER_IMAGE_L = 3260.0         # MQ-GLIP-L (Swin-L 1625 + 8 fusion layers ...), BASELINE.md section 2


def executed_gflop_per_image(n_tok, layers=6, swin=198.2):
    """FLOPs this implementation EXECUTES per image-forward (2 * MAC, MFMA + library GEMMs), as opposed to the reference's un-folded
    algorithmic 1451 GF (SURVEY.md 8d, T = 256):  Swin + FPN 29.3 + BERT 45.9 + pre-select 7.5 + GCP 17.9 (K / V projected once per
    unique vision token: 6 x 2.0 GF less) + per fusion layer {VLFuse: image-side projections folded into the text operands (70.5 GF gone;
    folded text GEMMs 2.4), QK^T computed by both directions over the live key blocks; BERT layer 3.8; DyConv 42.4} + heads over the live
    text blocks."""
    N = sum(h * w for h, w in LEVELS)
    k16, k32 = -(-n_tok // 16) * 16, -(-n_tok // 32) * 32
    vlfuse_attn = (2.0 * 8 * N * 256 * (k16 + k32) + 4.0 * 8 * N * 256 * k16) / 1e9
    per_layer = vlfuse_attn + 2.4 + 3.83 + 42.4
    heads = 2.0 * N * 256 * (k16 + 16) / 1e9
    return swin + 29.3 + 45.9 + 7.5 + 17.9 + layers * per_layer + heads


def build_model(dev, caches=False, n_classes=NUM_CLASSES_IN_CAPTION, n_categories=None, large=False, words=None, dtype="f16"):
    from transformers import AutoTokenizer
    from mq_det_amd import get_cfg
    from mq_det_amd.modeling.detector import GeneralizedVLRCNN_New
    from mq_det_amd.utils.synth import randomize_, synthetic_bank
    from mq_det_amd.utils.tokenizer import build_synthetic_tokenizer, synthetic_caption, positive_map_from_spans
    cfg = get_cfg()
    # LVIS-style evaluation settings of configs/vision_query_5shot/lvis_minival.yaml
    cfg.MODEL.DYHEAD.NUM_CLASSES = 1204
    cfg.MODEL.ATSS.DETECTIONS_PER_IMG = 300
    cfg.TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM = 3000
    cfg.MODEL.BACKBONE_CACHE = bool(caches)
    # operand type of the kernels (fp32 accumulate); f32 = the precise mode (fp32 operands: parity at 1e-3 on the device, not throughput)
    cfg.MODEL.COMPUTE_DTYPE = {"bf16": "bfloat16", "f32": "float32"}.get(dtype, "float16")
    if large:                                                 # configs/pretrain/mq-glip-l.yaml:11-17,41
        cfg.MODEL.SWINT.EMBED_DIM, cfg.MODEL.SWINT.DEPTHS = 192, (2, 2, 18, 2)
        cfg.MODEL.SWINT.NUM_HEADS, cfg.MODEL.SWINT.WINDOW_SIZE = (6, 12, 24, 48), 12
        cfg.MODEL.SWINT.OUT_CHANNELS = (192, 384, 768, 1536)
        cfg.MODEL.DYHEAD.NUM_CONVS = 8
    if os.environ.get("MQ_COMPACT_TEXT") is not None:         # A/B switch (live-row compaction of the text path, detector._live_len)
        cfg.MODEL.LANGUAGE_BACKBONE.COMPACT_TEXT = os.environ["MQ_COMPACT_TEXT"] == "1"
    if os.environ.get("MQ_RESIDUAL_FP32") is not None:        # A/B switch (precision of the residual streams)
        cfg.MODEL.RESIDUAL_FP32 = os.environ["MQ_RESIDUAL_FP32"] == "1"
    if os.environ.get("MQ_SWIN_FUSED_MLP") is not None:       # A/B switch (fused Swin MLP kernel vs library GEMMs + GELU)
        cfg.MODEL.SWINT.FUSED_MLP = os.environ["MQ_SWIN_FUSED_MLP"] == "1"
    if os.environ.get("MQ_SWIN_MLP_WIDTHS") is not None:      # A/B switch: which Swin stages use the fused MLP kernel
        cfg.MODEL.SWINT.FUSED_MLP_WIDTHS = tuple(int(w) for w in os.environ["MQ_SWIN_MLP_WIDTHS"].split(",") if w)
    tok_dir = build_synthetic_tokenizer(tempfile.mkdtemp(prefix="mqdet_tok_"))
    cfg.MODEL.LANGUAGE_BACKBONE.TOKENIZER_TYPE = tok_dir
    tk = AutoTokenizer.from_pretrained(tok_dir)
    model = GeneralizedVLRCNN_New(cfg, tokenizer=tk)
    randomize_(model, seed=0)
    chunks = []
    n_categories = n_categories or n_classes
    for c0 in range(0, n_categories, n_classes):            # chunk captions (engine/inference.py:190-192)
        n = min(n_classes, n_categories - c0)
        caption, spans = synthetic_caption(n, start=3 * c0, **({} if words is None else {"words": words}))
        chunks.append((caption, positive_map_from_spans(tk, caption, spans, list(range(c0 + 1, c0 + n + 1)))))
    model.load_query_bank(synthetic_bank(range(1, n_categories + 1), cfg.MODEL.BACKBONE.OUT_CHANNELS, cfg.VISION_QUERY.NUM_QUERY_PER_CLASS))
    model.to(dev)
    model.prepare(dev)
    return cfg, model, chunks


GFLOP_PER_IMAGE_GDINO = 1112.0     # Swin-T 198 + BERT 46 + GCP 37 + encoder 6 x 129 + two-stage 12 + decoder 6 x 7


def build_gdino_model(dev, n_classes=NUM_CLASSES_IN_CAPTION, dtype="f16"):
    """BASELINE.json configs[4]: MQ-GroundingDINO-T (configs/pretrain/mq-groundingdino-t.yaml), seeded random weights."""
    from mq_det_amd.config import get_gdino_cfg
    from mq_det_amd.modeling.detector import build_detection_model
    from mq_det_amd.utils.synth import randomize_, synthetic_bank
    from mq_det_amd.utils.tokenizer import build_synthetic_tokenizer, synthetic_caption, positive_map_from_spans
    cfg = get_gdino_cfg()
    cfg.MODEL.BACKBONE_CACHE = False          # every step a full forward (the synthetic loop re-sends the same tensor)
    cfg.MODEL.COMPUTE_DTYPE = {"bf16": "bfloat16", "f32": "float32"}.get(dtype, "float16")       # f32 = the split-precise mode (round 6: MQ-GroundingDINO too)
    cfg.GROUNDINGDINO.text_encoder_type = build_synthetic_tokenizer(tempfile.mkdtemp(prefix="mqdet_tok_"))
    model = build_detection_model(cfg)
    randomize_(model, seed=0)
    caption, spans = synthetic_caption(n_classes)
    pmap = positive_map_from_spans(model.tokenizer, caption + ".", spans, list(range(1, n_classes + 1)))
    model.load_query_bank(synthetic_bank(range(1, n_classes + 1), cfg.GROUNDINGDINO.hidden_dim, cfg.VISION_QUERY.NUM_QUERY_PER_CLASS))
    model.to(dev)
    model.prepare(dev)
    return cfg, model, caption, pmap


def main_gdino(args, rank, world, dev):
    """--workload mq-gdino-t: one step = one GroundingDINO.forward (Swin -> input projections -> BERT + GCP -> 6 encoder layers
    (fusion, text enhancer, deformable attention) -> two-stage selection -> 6 decoder layers -> heads -> BoxLists) for a batch of
    16 images 800 x 1333 per GPU; images shard across ranks, no data-path collective besides the gather of detections."""
    from mq_det_amd import ops, parallel
    from mq_det_amd.structures import ImageList
    cfg, model, caption, pmap = build_gdino_model(dev, dtype=args.dtype or "f16")
    if args.no_graph:
        model.use_hip_graph = False
    Bn = 16 if args.batch == B_PER_GPU else args.batch
    g = torch.Generator().manual_seed(1000 + rank)
    H, W = IMG_HW
    Hp, Wp = -(-H // 32) * 32, -(-W // 32) * 32
    imgs = torch.zeros(Bn, 3, Hp, Wp)
    imgs[:, :, :H, :W] = torch.randn(Bn, 3, H, W, generator=g)
    images = ImageList(imgs.to(dev), [(H, W)] * Bn)
    captions = [caption] * Bn
    # With seeded random weights no query reaches the yaml's box_threshold, and a step without survivors would not time the conversion /
    # gather of real rows: one probe forward with the threshold at 0, then the threshold = the MEDIAN best-class score
    # of the 900 queries -- about half of them survive, like a confident model on a crowded image.
    yaml_thr, graph_on = float(cfg.GROUNDINGDINO.box_threshold), model.use_hip_graph
    cfg.GROUNDINGDINO.box_threshold, model.use_hip_graph = 0.0, False
    model(ImageList(imgs[:2].to(dev), [(H, W)] * 2), captions=captions[:2], positive_map=pmap)
    sc = model.last_packed[..., 4]
    bench_thr = float(sc[sc > 0].median()) if bool((sc > 0).any()) else yaml_thr
    cfg.GROUNDINGDINO.box_threshold, model.use_hip_graph = bench_thr, graph_on
    model.clear_caches()

    def step():
        out = model(images, captions=captions, positive_map=pmap)
        if world > 1:
            parallel.gather_detections(model.last_packed)
        return out
    for _ in range(max(args.warmup, 2)):
        step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    kern, prof_steps = {}, 0
    if rank == 0:
        prof_steps = min(2, args.steps)
        ops.start_timing()
        for _ in range(prof_steps):
            model(images, captions=captions, positive_map=pmap)
        kern = ops.stop_timing()
        from mq_det_amd.modeling import gdino_pipeline as gp
        graph_on, model.use_hip_graph = model.use_hip_graph, False
        gp.start_marks()
        for _ in range(prof_steps):
            model(images, captions=captions, positive_map=pmap)
        stages = gp.stop_marks()
        model.use_hip_graph = graph_on
        ips = world * Bn * args.steps / dt
        res = {"metric": "images/sec MQ-GroundingDINO-T 800×1333 5-shot vision queries", "value": round(ips, 3), "unit": "images/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype or "f16", "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[4]: MQ-GroundingDINO-T (Swin-T + BERT-base + GCP + 6 + 6 deformable "
                                      "transformer layers, 900 queries), 5 vision queries x 40 classes, every step a full forward",
                          "global_batch": world * Bn, "batch_per_gpu": Bn, "image": "800x1333 -> 800x1344", "parallelism": f"dp{world}",
                          "weights": "seeded random init (no checkpoints offline)",
                          "box_threshold": f"{bench_thr:.4f} = median query score of a probe forward (the yaml's {yaml_thr} leaves no survivor with random weights)"},
               "detections_img0": len(out[0]), "hip_graph": bool(model.use_hip_graph and any(e.get("stage") == 2 for e in model._graphs.values())),
               "cache_stats": dict(model.cache_stats), "model_tflops": round(ips * GFLOP_PER_IMAGE_GDINO / 1e3, 2),
               "model_frac_of_mfma_peak": round(ips * GFLOP_PER_IMAGE_GDINO / 1e3 / (MFMA_PEAK_TFLOPS * world), 4),
               "kernels_ms_per_step": {k: round(v[1] / max(prof_steps, 1), 3) for k, v in sorted(kern.items())},
               "kernels_gbs": {k: round(v[2] / (v[1] * 1e-3) / 1e9, 1) for k, v in sorted(kern.items()) if v[2] and v[1] > 0},
               "stages_ms_per_step": {k: round(v / max(prof_steps, 1), 3) for k, v in stages.items()},
               "timing": "kernels_ms_per_step: HIP events around each hand-written launch in an eager pass of the same step; "
                         "stages_ms_per_step: HIP events between pipeline stages in an eager pass (host launch gaps included)"}
        msda = [(k, v) for k, v in kern.items() if k.startswith("msdeform_attn_q")]
        if msda:                                   # the gather kernel of the deformable attention: bound by 64-byte corner rows through L2
            k, v = max(msda, key=lambda kv: kv[1][1])
            gathered = Bn * int(k[len("msdeform_attn_q"):]) * 8 * 16 * 4 * 64.0 * v[0]
            res["roofline"] = {"bound": "hbm", "kernel": "msda_q_kernel (MSDeformAttn: softmax + sampling locations + bilinear gather), encoder shape",
                               "achieved": round(v[2] / (v[1] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(v[2] / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                               "avg_launch_ms": round(v[1] / v[0], 4), "l2_gather_GBs": round(gathered / (v[1] * 1e-3) / 1e9, 1),
                               "note": "achieved = ALGORITHMIC bytes (value once + projection + output) / time; l2_gather_GBs = bytes of the 64-byte "
                                       "corner rows the gather pulls through L2 -> L1 (the real bound, DESIGN.md 11)"}
        if world == 1 and args.cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(timeout=300, flag="--cpu-baseline-worker-gdino")
        print(compact_line(res, args.extras_file or os.path.join(ROOT, "bench_extras_gdino.json"), write=bool(args.extras_file) or not args.no_extras), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


CPU_BASELINE_THREADS = 32      # the oracle's many small torch ops stop scaling (and can crawl) far below 256 threads


def _cpu_baseline_worker():
    """BASELINE.md section 3 (configs[0]): the CPU oracle (pure-PyTorch fp32 restatement of the reference; the reference itself has no CPU
    path) on one 800x1333 image with a 20-token caption, TWICE: GLIP-T without vision queries (1 warm-up + 5 timed forwards -- `value`) and
    MQ-GLIP-T with a 5-shot query bank (pre-select + 6 GCP blocks on; 1 warm-up + 2 timed -- `mq_glip_t`); min and median of each (VERDICT
    r5 #9: the median alone wandered 0.15 ... 0.23 between boxes).  A bounded sample: ~60 s of CPU work; threads pinned by the parent
    (OMP_PLACES=cores, OMP_PROC_BIND=close)."""
    from oracle import glip_t_spec
    from oracle import detector as od
    from oracle.weights import make_state_dict, make_query_bank
    threads = min(os.cpu_count() or 1, CPU_BASELINE_THREADS)
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    images, sizes = od.pad_images([torch.randn(3, *IMG_HW, generator=g)], 32)
    pm = {1: [1], 2: [3], 3: [5, 6], 4: [8], 5: [10, 11], 6: [13]}
    t_all = time.time()
    out = {}
    for key, vq, timed in (("glip_t", False, 5), ("mq_glip_t", True, 2)):
        spec = glip_t_spec(vision_query=vq)
        sd = make_state_dict(spec, 0)
        T, nvalid = spec.max_query_len, 20
        ids = torch.zeros(1, T, dtype=torch.long)
        ids[:, :nvalid] = torch.randint(1000, spec.vocab, (nvalid,), generator=g)
        am = torch.zeros(1, T, dtype=torch.long)
        am[:, :nvalid] = 1
        bank = make_query_bank(pm.keys(), spec) if vq else None
        times = []
        with torch.no_grad():
            for i in range(1 + timed):
                t = time.time()
                od.forward(sd, spec, images, sizes, ids, am, pm, bank)
                if i >= 1:
                    times.append(time.time() - t)
        out[key] = {"value": round(1.0 / statistics.median(times), 4), "best": round(1.0 / min(times), 4),
                    "median_s_per_forward": round(statistics.median(times), 3), "min_s_per_forward": round(min(times), 3), "forwards": timed}
        del sd
    a = out["glip_t"]
    print(json.dumps({"value": a["value"], "unit": "images/sec", "cores": threads, "kind": "port",
                      "median_s_per_forward": a["median_s_per_forward"], "min_s_per_forward": a["min_s_per_forward"],
                      "mq_glip_t": out["mq_glip_t"],
                      "why_not_all_cores": "BASELINE.md section 3 asks for all host cores; with the 256 threads of the GPU box the oracle's many small "
                                           "torch ops crawl -- an all-core sample did not finish inside 6 minutes (GPU call 14 of round 4) -- so the "
                                           "thread count is capped and stated",
                      "sample": f"one 800x1333 image (padded 800x1344), 20-token caption, fp32 CPU oracle: GLIP-T (no vision queries) 1 warm-up + 5 timed "
                                f"forwards = value (median); mq_glip_t = the same with a 5-shot query bank, 1 + 2 forwards; {time.time() - t_all:.0f} s in "
                                f"total, {threads} pinned torch threads on a {os.cpu_count()}-core host"}), flush=True)


def _cpu_baseline_worker_gdino():
    """The same for --workload mq-gdino-t: the CPU oracle of MQ-GroundingDINO-T (oracle/gdino.py, pinned to the reference's own
    module) WITHOUT vision queries on one 800x1333 image with a 20-token caption; 1 warm-up + 3 timed forwards, median."""
    from oracle import gdino as og
    from oracle.spec import gdino_t_spec
    from oracle.weights import make_gdino_state_dict
    threads = min(os.cpu_count() or 1, CPU_BASELINE_THREADS)
    torch.set_num_threads(threads)
    spec = gdino_t_spec(vocab=30522, vision_query=False)
    sd = make_gdino_state_dict(spec, 0)
    g = torch.Generator().manual_seed(0)
    H, W = IMG_HW
    img = torch.zeros(1, 3, -(-H // 32) * 32, -(-W // 32) * 32)
    img[:, :, :H, :W] = torch.randn(1, 3, H, W, generator=g)
    ids = torch.zeros(1, 512, dtype=torch.long)
    row = [101] + [2000 + i if (i + 1) % 3 else 1012 for i in range(18)] + [102]           # [CLS] w w . w w . ... [SEP]
    ids[0, :len(row)] = torch.tensor(row)
    am = (ids != 0).long()
    pm = {k + 1: [1 + 3 * k, 2 + 3 * k] for k in range(6)}
    times = []
    t_all = time.time()
    with torch.no_grad():
        for i in range(4):
            t = time.time()
            og.forward(sd, spec, img, [(H, W)], ids, am, pm, [101, 102, 1012, 1029])
            if i >= 1:
                times.append(time.time() - t)
    med = statistics.median(times)
    print(json.dumps({"value": round(1.0 / med, 4), "unit": "images/sec", "cores": threads, "kind": "port",
                      "median_s_per_forward": round(med, 3),
                      "sample": f"MQ-GroundingDINO-T (no vision queries), one 800x1333 image (padded 800x1344), 20-token caption; 1 warm-up "
                                f"+ 3 timed forwards of the fp32 CPU oracle, median; {time.time() - t_all:.1f} s in total, {threads} torch "
                                f"threads on a {os.cpu_count()}-core host"}), flush=True)


def cpu_baseline_start(flag="--cpu-baseline-worker"):
    """Start the CPU-oracle worker (its own process, pinned threads, no GPU visible)."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(CPU_BASELINE_THREADS), MKL_NUM_THREADS=str(CPU_BASELINE_THREADS),
               OMP_PLACES="cores", OMP_PROC_BIND="close", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    return subprocess.Popen([sys.executable, os.path.abspath(__file__), flag], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def cpu_baseline_finish(proc, timeout=420):
    import subprocess
    try:
        out, err = proc.communicate(timeout=timeout)
        for line in reversed(out.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": (err or out)[-300:]}
    except subprocess.TimeoutExpired:
        proc.kill()
        return {"error": f"cpu baseline worker exceeded {timeout} s", "cores": CPU_BASELINE_THREADS, "kind": "port"}


def cpu_baseline(timeout=420, flag="--cpu-baseline-worker"):
    """Run the worker in a subprocess with a hard time limit so the default bench run stays bounded."""
    return cpu_baseline_finish(cpu_baseline_start(flag), timeout)


def _mfma(kernel, flops_alg, flops_exec, n_launch, ms, note, traffic=None):
    if _PMC_SPLIT:                      # split-precise: every algorithmic MFMA is executed as three fp16 MFMAs (hi hi + hi lo + lo hi)
        flops_exec, note = 3.0 * flops_exec, note + "; SPLIT-PRECISE: executed = 3 x (three fp16 MFMAs per contraction), achieved / frac count the ALGORITHMIC flops"
    ach = flops_alg / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": kernel, "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "avg_launch_ms": round(ms / max(n_launch, 1), 4),
            "launches_per_step": n_launch, "ms_per_step": round(ms, 3),
            "flops_per_step": {"algorithmic": flops_alg, "executed": flops_exec},
            "executed_tflops": round(flops_exec / (ms * 1e-3) / 1e12, 2), "note": note}


def _hbm(kernel, nbytes, n_launch, ms, note):
    ach = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_ms": round(ms / max(n_launch, 1), 4),
            "launches_per_step": n_launch, "ms_per_step": round(ms, 3), "algorithmic_bytes_per_step": nbytes, "note": note}


_PMC_SPLIT = False       # main() sets it for --dtype f32: the split-precise kernels have their own counter file


def _pmc():
    """PMC traffic (HBM bytes per launch) collected by tools/pmc_traffic.sh in separate rocprofv3 passes, reduced by
    tools/pmc_reduce.py; the newest round's file first, round 1 (VLFuse only) as fallback.  Split-precise runs read r0N_pmc_traffic_split.json."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_traffic_split.json" if _PMC_SPLIT else "r0*_pmc_traffic.json")), reverse=True)      # newest round first
    if _PMC_SPLIT and not files:
        return {}
    if files:
        d = json.load(open(files[0]))
        out = {k: int(v["traffic_bytes"]) for k, v in d.get("kernels", {}).items()}
        out["vlfuse"] = d.get("vlfuse_traffic_bytes_per_launch_avg")
        out["_file"] = os.path.relpath(files[0], ROOT)
        return out
    r1 = os.path.join(ROOT, "profiles", "r01_pmc_vlfuse.json")
    if os.path.exists(r1):
        return {"vlfuse": json.load(open(r1)).get("traffic_bytes_per_launch_avg"), "_file": "profiles/r01_pmc_vlfuse.json"}
    return {}


def kernel_rooflines(kern, steps, Bn, n_tok, embed=96):
    """kern: ops.stop_timing() of `steps` eager single-stream forwards -> per-kernel roofline records (per step)."""
    pmc = _pmc()
    N_img = sum(h * w for h, w in LEVELS)
    per = {k: (v[0] / steps, v[1] / steps, v[2] / steps) for k, v in kern.items()}
    out = []
    # ---- DCNv2 (grouped launch, dcn_fused.hip).  Algorithmic = executed: 2 * positions * (9 * 256) * 256 per DyConv layer,
    # positions = 33600 / image (same-level + stride-2 + level+1 branches, vldyhead.py:205-247) -- SURVEY.md 8(d) 42.4 GF/layer
    if "dcnv2_fused" in per:
        n, ms, _ = per["dcnv2_fused"]
        pos = sum(h * w for h, w in LEVELS) + 2 * sum(h * w for h, w in LEVELS[1:])
        fl = n * 2.0 * pos * Bn * 2304 * 256
        out.append(_mfma("dcn_igemm8_kernel (DCNv2: bilinear gather + blend + MFMA + GroupNorm statistics, 13 branches of a DyConv "
                         "layer per launch)", fl, fl, n, ms, "flops = 2 * 33600 * B * 2304 * 256 per layer (SURVEY.md 8d: 42.4 GF / image / layer); "
                         "traffic: PMC bytes per launch (profiles/r0N_pmc_traffic.json, newest round)", pmc.get("dcn_igemm8_kernel")))
    if "dcnv2_fpn" in per:                       # the FPN output convs (3 levels, one grouped launch) + P6 / P7 through the same kernel
        n, ms, _ = per["dcnv2_fpn"]
        pos = sum(h * w for h, w in LEVELS)      # 16800 + 4200 + 1050 (stride 1) + 273 + 77 (stride 2) output positions
        fl = 2.0 * pos * Bn * 2304 * 256
        out.append(_mfma("dcn_igemm8_kernel as plain 3x3 conv (FPN output convs, zero offsets)", fl, fl, n, ms,
                         "flops = 2 * 22400 * B * 2304 * 256 (fpn_layer2-4 + P6 + P7); 3 launches"))
    # ---- VLFuse attention (vlfuse_attn.hip), 8 heads x 256.  SURVEY.md 8(d) / fuse_helper.py:233 compute QK^T ONCE and two
    # PV products: algorithmic = 3 * 2 * B * 8 * N * T_vis * 256 per layer with T_vis = the 64-key tiles that hold caption
    # tokens.  The two kernels each recompute the logits (executed = 4 * ...); the text side only computes 128-row query tiles.
    i2t = [v for k, v in per.items() if k.startswith("vlfuse_i2t_n%d_" % N_img)]
    t2i = [v for k, v in per.items() if k.startswith("vlfuse_t2i_n%d_" % N_img)]
    if i2t and t2i:
        k16, k32, r16 = -(-n_tok // 16) * 16, -(-n_tok // 32) * 32, -(-n_tok // 16) * 16
        n_l = i2t[0][0]
        # executed: image side = QK^T over the live 16-key blocks + PV over the live 32-key steps; text side recomputes QK^T
        # and does its PV for the 16-row wave blocks that hold caption tokens
        ex_i = n_l * 2.0 * Bn * 8 * N_img * 256 * (k16 + k32)
        ex_t = t2i[0][0] * 4.0 * Bn * 8 * N_img * 256 * r16
        unit = 2.0 * Bn * 8 * N_img * 256 * n_tok                      # one contraction over the caption tokens
        # ONE record per kernel (the dominant-kernel choice compares single kernels).  The algorithmic QK^T (computed once in the reference,
        # fuse_helper.py:233) is booked on the image side, the text side is credited with its PV only.
        out.append(_mfma("vlfuse_i2t_kernel (VLFuse image->text attention: QK^T + softmax over text + PV, projections folded)", n_l * 2 * unit, ex_i, n_l,
                         i2t[0][1], f"algorithmic = QK^T + PV over the {n_tok} caption tokens; executed over {k16} / {k32} keys; traffic: PMC bytes per launch",
                         pmc.get("vlfuse_i2t_kernel")))
        tt = (pmc.get("vlfuse_t2i_kernel") or 0) + (pmc.get("vlfuse_t2i_combine_kernel") or 0)
        out.append(_mfma("vlfuse_t2i_kernel (+ merge) (VLFuse text->image attention: softmax over image tokens + PV)", t2i[0][0] * unit, ex_t, t2i[0][0],
                         t2i[0][1], f"algorithmic = PV only (the logits are the image side's, computed once in the reference); executed = QK^T again + PV "
                         f"for {r16} query rows; traffic: PMC bytes per launch incl. the merge of the key-split partials", tt or None))
        pair = _mfma("vlfuse_i2t_kernel + vlfuse_t2i_kernel (the pair; continuity with rounds 1-3)", 3 * n_l * unit, ex_i + ex_t, n_l + t2i[0][0],
                     i2t[0][1] + t2i[0][1], "algorithmic = QK^T once + 2 PV (SURVEY.md 8d); two kernels: not a candidate for `roofline`", pmc.get("vlfuse"))
        pair["pair"] = True
        out.append(pair)
    # ---- generic attention kernel (BERT self-attention 12 x 64; GCP pre-select 8 x 32): QK^T + PV over the visited keys
    att = [(k, v) for k, v in per.items() if k.startswith(("attn_d", "attn_res_d", "attn_text_d", "attn_chk_d"))]
    if att:
        fl = 0.0
        for k, (n, ms, _) in att:
            f = [x for x in k.split("_") if x[:1] == "d" or x[:2] in ("nq", "nk")]
            d, nq, nk = int(f[0][1:]), int(f[1][2:]), int(f[2][2:])
            heads = 12 if d == 64 else 8
            nk_eff = min(nk, -(-n_tok // 64) * 64) if d == 64 else nk
            fl += n * 4.0 * Bn * heads * nq * nk_eff * d
        out.append(_mfma("attn_text_kernel / attn_chunked_kernel (BERT self-attention 12 x 64, T = 256; GCP pre-select 8 x 32 over 5577 "
                         "image tokens)", fl, fl, sum(v[0] for _, v in att), sum(v[1] for _, v in att), "flops = 4 * B * heads * Nq * Nk_visited * D"))
    # ---- round 5: the attention half of a BERT layer as one launch (bert_attn.hip): q | k | v projection + attention on the live text rows
    ba = [(k, v) for k, v in per.items() if k.startswith("bert_attn_qkv_t")]
    if ba:
        fl = 0.0
        for k, (n, ms, _) in ba:
            T = int(k[len("bert_attn_qkv_t"):])
            k16 = min(T, -(-n_tok // 16) * 16)
            fl += n * (2.0 * Bn * T * 768 * 2304 + 4.0 * Bn * 12 * T * k16 * 64)
        out.append(_mfma("bert_attn_qkv_kernel (BertSelfAttention in one launch: q | k | v projection + attention per (batch item, head); 18 launches per step)",
                         fl, fl, sum(v[0] for _, v in ba), sum(v[1] for _, v in ba),
                         "flops = 2 * B * T * 768 * 2304 (projection) + 4 * B * 12 * T * keys * 64 (QK^T, PV) per launch, T = live text rows"))
    ga = [(k, v) for k, v in per.items() if k.startswith("gcp_attn_fused")]
    if ga:
        T = next((int(k[len("bert_attn_qkv_t"):]) for k, _ in ba), -(-n_tok // 16) * 16)
        fl = sum(v[0] for _, v in ga) * 2.0 * Bn * T * (768 * 512 + 512 * 768 + 768 * 384)
        out.append(_mfma("gcp_attn_kernel (GCP attention half in one launch: LayerNorm, to_q, sparse gather-attention, to_out, gate MLP, gated residual, "
                         "next LayerNorm; weights streamed L2 -> registers in MFMA B-fragment order)", fl, fl, sum(v[0] for _, v in ga), sum(v[1] for _, v in ga),
                         "flops = 2 * B * T * (768*512 + 512*768 + 768*384) per launch (the three projections; the <= 8-slot attention is negligible)"))
    # ---- fused Swin MLP (swin_mlp2.hip): 16 M C^2 per launch (fc1 + fc2), M = tokens of the stage
    mlp = [(k, v) for k, v in per.items() if k.startswith("swin_mlp_c")]
    if mlp:
        fl = 0.0
        for k, (n, ms, nbytes) in mlp:
            C = int(k[len("swin_mlp_c"):])
            M = Bn * (IMG_HW[0] // 4) * (-(-IMG_HW[1] // 32) * 32 // 4) * (embed / C) ** 2       # tokens of the stage whose width is C
            fl += n * 16.0 * M * C * C
        out.append(_mfma("swin_mlp2_kernel (LN + fc1 + exact GELU + fc2 + residual + next LN in one kernel; VALU (GELU) bound at C = 96)", fl, fl,
                         sum(v[0] for _, v in mlp), sum(v[1] for _, v in mlp), "flops = 16 * M * C^2 per launch (fc1 + fc2, hidden 4C)"))
    # ---- window attention with the qkv projection inside (window_attn.hip): projection 6 M C^2 + attention 4 M N C per launch
    wq = [(k, v) for k, v in per.items() if k.startswith("window_attn_qkv_c")]
    if wq:
        fl = fe = 0.0
        for k, (n, ms, nbytes) in wq:
            C = int(k[len("window_attn_qkv_c"):])
            M = Bn * (IMG_HW[0] // 4) * (-(-IMG_HW[1] // 32) * 32 // 4) * (embed / C) ** 2
            fl += n * (6.0 * M * C * C + 4.0 * M * 49 * C)
            fe += n * (6.0 * M * C * C + 4.0 * M * 64 * C) * 64 / 49            # 49-token windows run as 64 padded tokens
        out.append(_mfma("window_attn_qkv_kernel (Swin qkv projection + W-MSA / SW-MSA in one kernel; the qkv tensor is never written)", fl, fe,
                         sum(v[0] for _, v in wq), sum(v[1] for _, v in wq),
                         "flops = 6 * M * C^2 (qkv) + 4 * M * 49 * C (QK^T, PV) per launch; HBM bytes 4 * M * C (x in, out out)"))
    # ---- HBM-bound kernels: algorithmic bytes (every input read once, every output written once) / time
    groups = (("layernorm_kernel (all LayerNorms, residual add fused)", "layernorm_c"), ("window_attn_kernel (Swin W-MSA / SW-MSA)", "window_attn_c"),
              ("dyconv_fuse_kernel (GroupNorm affine + up-sampling + scale attention + branch mean)", "dyconv_fuse"),
              ("dyconv_fuse_group_kernel + dyrelu_coef_group_kernel (the same epilogue, all levels of a layer in two launches)", "dyconv_epilogue_group"),
              ("dyrelu_apply_kernel", "dyrelu_apply"), ("conv3x3_small_kernel (27-channel DyConv offset conv)", "conv3x3_small"),
              ("conv3x3_group_kernel (27-channel DyConv offset conv, all levels of a layer in one launch, weights in registers)", "conv3x3_group"),
              ("align_scores_kernel (sigmoid + token->class mean + threshold)", "align_scores"),
              ("align_fused_kernel (box / centerness heads + dot-product alignment + sigmoid + class aggregation + threshold, all levels)", "align_fused"),
              ("patch_embed_kernel (Swin PatchEmbed projection + patch_embed.norm + first norm1, fp32 NCHW pixels in)", "patch_embed_c"),
              ("post_select_kernel (exact top-1000 per level: radix select over the score maps, slices then levels, + box decode)", "post_select"))
    for name, prefix in groups:
        sel = [v for k, v in per.items() if k.startswith(prefix)]
        if sel:
            out.append(_hbm(name, sum(v[2] for v in sel), sum(v[0] for v in sel), sum(v[1] for v in sel),
                            "algorithmic bytes = inputs read once + outputs written once"))
    return out


def _lang_path_once(model, cfg, dev, chunks, iters=5):
    """North-star target line (BASELINE.json: ">= 40 % MFMA utilisation on the fused GCP+BERT attention at bs = 64 / GPU"):
    the language path -- embeddings, 12 BERT layers, GCP pre-select (2 layers over the 5577 pooled image tokens) and the
    6 gated cross-attention blocks -- at B = 64 on ONE GPU, eager single stream, HIP events.  Reports the whole path and,
    separately, its attention kernels (BERT 12 x 64, pre-select 8 x 32, GCP sparse) with the flop formula used."""
    from mq_det_amd import ops
    from mq_det_amd.modeling import pipeline
    from mq_det_amd.modeling.query_selector import build_token_index  # noqa: F401
    Bn = 64
    P = model._plan
    caption, pmap = chunks[0]
    ids, am, max_kv = model.tokenize([caption] * Bn, dev)
    n_tok = int(am[0].sum())
    labels = [k for k, v in pmap.items() if len(v)]
    pm_key = tuple((k, tuple(pmap[k])) for k in labels)
    dtype = P["backbone.body.patch_embed.proj.weight"].dtype
    Tl = model._live_len(ids.shape[1], max_kv)            # live-row compaction, exactly as forward() applies it
    ids, am = ids[:, :Tl].contiguous(), am[:, :Tl].contiguous()
    vision, idx = model.query_selector.select_cached(pm_key, labels, pmap, Bn, Tl, dev, dtype)
    g = torch.Generator(device="cpu").manual_seed(3)
    pooled = torch.randn(Bn, 5577, 256, generator=g).to(dev, dtype)
    streams_on = cfg.MODEL.DYHEAD.LEVEL_STREAMS
    cfg.MODEL.DYHEAD.LEVEL_STREAMS = False
    try:
        for _ in range(2):
            pipeline.language_backbone(P, cfg, ids, am, vision, pooled, idx, max_kv=max_kv)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            pipeline.language_backbone(P, cfg, ids, am, vision, pooled, idx, max_kv=max_kv)
        e1.record()
        torch.cuda.synchronize()
        ms_path = e0.elapsed_time(e1) / iters
        ops.start_timing()
        for _ in range(iters):
            pipeline.language_backbone(P, cfg, ids, am, vision, pooled, idx, max_kv=max_kv)
        kern = ops.stop_timing()
    finally:
        cfg.MODEL.DYHEAD.LEVEL_STREAMS = streams_on
    k16 = -(-n_tok // 16) * 16                            # text keys the attention kernels visit (16-key blocks)
    V, S = vision.shape[1], idx.shape[2]
    C = 768
    fused = any(k.startswith("bert_attn_qkv") for k in kern)
    # ALGORITHMIC flops of the launches counted below.  BERT: with mq_bert_attn_qkv_fwd one launch IS the layer's q | k | v projection and its
    # attention (rows = the Tl live text positions); without it only the attention launch is counted (its projection is a library GEMM).
    fl_bert_attn = 12 * 4.0 * Bn * 12 * Tl * k16 * 64
    fl_bert_proj = 12 * 2.0 * Bn * Tl * C * 3 * C if fused else 0.0
    fl_pre = 2 * 4.0 * Bn * 8 * V * 5577 * 32
    gcp_fused = any(k.startswith("gcp_attn_fused") for k in kern)
    # GCP: with mq_gcp_attn_fwd one launch is to_q + the sparse attention + to_out + the gate MLP (+ three LayerNorms, not counted)
    fl_gcp = 6 * 4.0 * Bn * 8 * Tl * S * 64 + (6 * 2.0 * Bn * Tl * (C * 512 + 512 * C + C * 384) if gcp_fused else 0.0)
    tags = ("attn_d", "attn_res_d", "attn_text_d", "attn_chk_d", "gcp_sparse", "bert_attn_qkv", "gcp_attn_fused")
    att_ms = sum(v[1] for k, v in kern.items() if k.startswith(tags)) / iters
    bert_ms = sum(v[1] for k, v in kern.items() if k.startswith(("bert_attn_qkv", "attn_text_d", "attn_res_d"))) / iters
    att_tf = (fl_bert_attn + fl_bert_proj + fl_pre + fl_gcp) / (att_ms * 1e-3) / 1e12
    bert_tf = (fl_bert_attn + fl_bert_proj) / max(bert_ms * 1e-3, 1e-9) / 1e12
    # whole language path: SURVEY.md 8(d) per image BERT 45.9 + pre-select 7.5 + GCP 29.9 GF at T = 256; the text GEMMs run on Tl rows now,
    # so the EXECUTED count scales the text-row share (everything but the pre-select and the K / V projections of the vision tokens) by Tl / 256
    path_gf_ref = 45.9 + 7.5 + 29.9
    path_gf_exec = (45.9 + 17.9) * Tl / 256.0 + 7.5
    path_tf = Bn * path_gf_exec * 1e9 / (ms_path * 1e-3) / 1e12
    return {"batch": Bn, "text_rows": Tl, "ms_language_path": round(ms_path, 3), "language_path_tflops": round(path_tf, 1),
            "language_path_frac_of_mfma_peak": round(path_tf / MFMA_PEAK_TFLOPS, 4),
            "language_path_gflop_per_image": {"executed": round(path_gf_exec, 1), "reference_at_T256": path_gf_ref},
            "attention_kernels_ms": round(att_ms, 3), "attention_tflops": round(att_tf, 1),
            "attention_mfma_utilisation": round(att_tf / MFMA_PEAK_TFLOPS, 4), "target": 0.40,
            "bert_fused_launches": {"ms": round(bert_ms, 3), "tflops": round(bert_tf, 1), "mfma_utilisation": round(bert_tf / MFMA_PEAK_TFLOPS, 4),
                                    "projection_inside_the_launch": fused},
            "flops": {"bert_self_attention": fl_bert_attn, "bert_qkv_projection_inside_the_attention_launch": fl_bert_proj,
                      "gcp_pre_select": fl_pre, "gcp_attention_launches": fl_gcp, "gcp_projections_inside_the_launch": gcp_fused,
                      "formula": f"12 x (4*B*12*{Tl}*{k16}*64" + (f" + 2*B*{Tl}*768*2304" if fused else "") + f") + 2 x 4*B*8*{V}*5577*32 + 6 x (4*B*8*{Tl}*{S}*64" + (f" + 2*B*{Tl}*(768*512 + 512*768 + 768*384)" if gcp_fused else "") + "), "
                                 f"B = {Bn} ({Tl} = live text rows after compaction, {k16} = visited text keys of the {n_tok}-token caption)"},
            "kernels_ms": {k: round(v[1] / iters, 3) for k, v in sorted(kern.items())},
            "note": "flops are counted on the LIVE text rows (round 5 runs the text on 16 ceil(live / 16) positions; rounds 1-4 ran and counted all 256 "
                    "padded rows, i.e. 1.78 x these attention flops for the 141-token caption): the utilisation of this record is not comparable "
                    "with earlier rounds' -- ms_language_path is",
            "timing": "eager, single stream, HIP events; attention = the bert_attn_qkv (projection + attention) / gcp_attn_fused (projections + sparse "
                      "attention + gate) / attn_* / gcp_sparse launches only"}


def lang_path_b64(model, cfg, dev, chunks, iters=5):
    """The north-star target line at B = 64, twice: under the DEFAULT kernel policy (since the weights of the two fused text kernels are
    streamed in MFMA B-fragment order -- GPU call 17 of round 5 -- the size rules of KERNELS["BERT_ATTN_QKV_FUSED"] / ["GCP_ATTN_FUSED"] = 1 take
    them at this batch: the north-star's "fused GCP + BERT attention") and with both OFF (= 0: the library GEMMs + the round-4 attention
    launches) for comparison."""
    from mq_det_amd import ops
    res = _lang_path_once(model, cfg, dev, chunks, iters)
    saved = (ops.KERNELS["BERT_ATTN_QKV_FUSED"], ops.KERNELS["GCP_ATTN_FUSED"])
    try:
        ops.KERNELS["BERT_ATTN_QKV_FUSED"], ops.KERNELS["GCP_ATTN_FUSED"] = 0, 0
        forced = _lang_path_once(model, cfg, dev, chunks, iters)
    finally:
        ops.KERNELS["BERT_ATTN_QKV_FUSED"], ops.KERNELS["GCP_ATTN_FUSED"] = saved
    res["fused_text_kernels_off"] = {k: forced[k] for k in ("ms_language_path", "language_path_frac_of_mfma_peak", "attention_kernels_ms", "attention_tflops",
                                                                 "attention_mfma_utilisation", "bert_fused_launches", "flops", "kernels_ms")}
    return res


def _sub_bench(argv, env=None, timeout=150, keep=()):
    """One more invocation of this script in a subprocess with a hard time limit (so that nothing it does can cost the headline line);
    returns a summary of the JSON line it printed."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), *argv, "--no-extras"], env=dict(os.environ, **(env or {})),
                           capture_output=True, text=True, timeout=timeout)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                d = json.loads(line)
                out = {k: d.get(k) for k in ("metric", "value", "unit", "ms_per_step", "dtype", "steps", "hip_graph", "detections_img0") + tuple(keep)
                       if d.get(k) is not None}
                out["workload"] = d.get("config", {}).get("workload")
                out["batch_per_gpu"] = d.get("config", {}).get("batch_per_gpu")
                return out
        return {"error": (r.stderr or r.stdout)[-400:]}
    except subprocess.TimeoutExpired:
        return {"error": f"worker exceeded {timeout} s"}


# every operator with two implementations on the one that was the default at the end of ROUND 3 (ops.KERNEL_DEFAULTS lists today's), and the
# runtime default of the hardware queues: what round 4's switchable changes buy (not switchable, so inside both runs: the VLFuse softmax diet)
ROUND3_KERNEL_SET = {"MQ_POST_FUSED": "0", "MQ_BERT_QKV_FUSED": "0", "MQ_PATCH_EMBED_FUSED": "0", "MQ_OFFSET_CONV_VARIANT": "2", "GPU_MAX_HW_QUEUES": "4"}
# ... and what ROUND 5's switchable changes buy: the text on all 256 padded positions, the BERT / GCP attention halves as the launches of round 4,
# the FPN convs through the general DCNv2 instantiation, the DyConv epilogue per level on five streams (not switchable, so inside both runs: the
# Swin MLP tail kernel with its loads in flight)
ROUND4_KERNEL_SET = {"MQ_COMPACT_TEXT": "0", "MQ_BERT_ATTN_QKV_FUSED": "0", "MQ_GCP_ATTN_FUSED": "0", "MQ_DCN_PLAIN": "0", "MQ_DYCONV_EPILOGUE_GROUPED": "0"}


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _launch_ranks(n):
    """Re-run this command line under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1, a free port unless
    MASTER_PORT is set) and exit with its status: `python bench.py --gpus 8` and the driver's explicit torchrun form run the same ranks."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    print(f"bench.py: starting {n} ranks: {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def _comm_info(world):
    """What the ranks talk over: backend, RCCL version (torch reports it as nccl), rank count."""
    import torch.distributed as dist
    info = {"ranks": world, "backend": dist.get_backend() if dist.is_initialized() else None}
    try:
        if torch.cuda.is_available():
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:  # noqa: BLE001
        info["rccl_version"] = f"unavailable ({type(e).__name__})"
    return info


def _dry_launch(args, rank, local, world):
    """Launcher check without kernels: every rank contributes a [B, K, 6] block of dummy detections to the one fixed-shape gather of the
    data path, rank 0 prints a line with the contract's launch fields.  Runs on CPU (gloo) or on GPUs (RCCL)."""
    import torch.distributed as dist
    from mq_det_amd import parallel
    dev = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
    Bn, K = args.batch, 316
    packed = torch.full((Bn, K, 6), float(rank + 1), device=dev)
    t0 = time.perf_counter()
    if args.overlap_gather:                              # the one-step-behind form: every submit but the first returns the previous block
        og, got = parallel.OverlappedGather(), 0
        for _ in range(max(args.steps, 1)):
            got += og.submit(packed) is not None
        allp = og.flush()
        assert got == max(args.steps, 1) - 1
    else:
        for _ in range(max(args.steps, 1)):
            allp = parallel.gather_detections(packed)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ok = allp.shape == (world * Bn, K, 6) and all(float(allp[r * Bn, 0, 0]) == r + 1 for r in range(world))
    if rank == 0:
        print(json.dumps({"metric": "images/sec MQ-GLIP-T 800×1333 5-shot vision queries, 1/2/4/8 MI355X", "value": None, "unit": "images/sec",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_launch": True, "gather_ok": bool(ok), "overlap_gather": bool(args.overlap_gather),
                          "gather_ms": round(1e3 * dt / max(args.steps, 1), 3), "comm": _comm_info(world),
                          "config": {"workload": "launcher check only: no forward", "global_batch": world * Bn, "batch_per_gpu": Bn,
                                     "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lang-b64", action="store_true")
    ap.add_argument("--batch", type=int, default=B_PER_GPU)
    ap.add_argument("--workload", choices=["mq-glip-t", "lvis", "mq-glip-l", "mq-gdino-t"], default="mq-glip-t")
    ap.add_argument("--caption", choices=["lvis", "short"], default="lvis", help="lvis: 1-4 word class names, 141 tokens (default); "
                                                                                  "short: one token per class, 81 tokens (the round-1 caption)")
    ap.add_argument("--chunk-batch", type=int, default=0, help="lvis workload: image x chunk items stacked per launch sequence "
                                                                "through model.forward_chunks (0 = one forward per chunk)")
    ap.add_argument("--dtype", choices=["f16", "bf16", "f32"], default=None, help="16-bit operand type of the kernels: f16 (default; BASELINE "
                                                                          "configs[1]) or bf16 (default of --workload mq-glip-l, configs[3])")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay (for PMC profiling)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-worker-gdino", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-experimental", action="store_true", help="skip the kernel-set A/B and the other BASELINE configs (subprocesses)")
    ap.add_argument("--extras-file", default=None, help="write the full record (rooflines, kernels_ms_per_step, sub-runs ...) to this path; default: "
                                                         "bench_extras.json next to this script, nothing under --no-extras")
    ap.add_argument("--no-extras", action="store_true", help="the contract fields + rooflines only (what the subprocess lines use)")
    ap.add_argument("--cpu-baseline", action="store_true", help="mq-gdino-t workload: also time the CPU oracle (off by default there)")
    ap.add_argument("--overlap-gather", action="store_true", help="N > 1: the gather of step k runs asynchronously under the forward of step k + 1 "
                                                                  "(parallel.OverlappedGather); the last one is waited for inside the timed region")
    ap.add_argument("--dry-launch", action="store_true", help="launcher check: start the ranks, rendezvous, one fixed-shape gather of dummy "
                                                              "detections, print the line with n_gpus = ranks; no kernels (runs on CPU over gloo)")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return _cpu_baseline_worker()
    if args.cpu_baseline_worker_gdino:
        return _cpu_baseline_worker_gdino()
    if args.no_extras:
        args.no_experimental = args.no_lang_b64 = args.no_cpu_baseline = True
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher around it: start the N ranks ourselves, one process per GPU
        # (the reference: `python -m torch.distributed.launch --nproc_per_node=N tools/test_grounding_net.py`, :35-60)
        return _launch_ranks(args.gpus)

    from mq_det_amd import parallel
    from mq_det_amd import ops
    from mq_det_amd.structures import ImageList
    t_start = time.perf_counter()
    rank, local, world = parallel.init_distributed()
    if world != args.gpus:
        # never report a number for a rank count other than the one that was asked for
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: refusing to run", file=sys.stderr, flush=True)
        raise SystemExit(3)
    if args.dry_launch:
        return _dry_launch(args, rank, local, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: rank {rank} (local {local}) has no GPU: {torch.cuda.device_count()} visible, --gpus {args.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ops.load_library()
    if args.workload == "mq-gdino-t":
        return main_gdino(args, rank, world, dev)
    lvis, large = args.workload == "lvis", args.workload == "mq-glip-l"
    if large and args.batch == B_PER_GPU:
        args.batch = 4                                        # BASELINE.json configs[3]: bs = 4 / GPU
    if args.dtype is None:
        args.dtype = "bf16" if large else "f16"               # BASELINE.json configs[3]: "MQ-GLIP-L ... bf16 MFMA"
    cfg, model, chunks = build_model(dev, caches=lvis, n_categories=1203 if lvis else None, large=large,
                                     words=(1,) if args.caption == "short" else None, dtype=args.dtype)
    if args.no_graph:
        model.use_hip_graph = False
    global _PMC_SPLIT
    _PMC_SPLIT = args.dtype == "f32"

    Bn = args.batch
    g = torch.Generator().manual_seed(1000 + rank)
    H, W = IMG_HW
    Hp, Wp = -(-H // 32) * 32, -(-W // 32) * 32
    imgs = torch.zeros(Bn, 3, Hp, Wp)
    imgs[:, :, :H, :W] = torch.randn(Bn, 3, H, W, generator=g)
    imgs = imgs.to(dev)
    images = ImageList(imgs, [(H, W)] * Bn)
    caption, pmap = chunks[0]
    captions = [caption] * Bn
    n_tok = int(model.tokenize(captions, dev)[1][0].sum())

    if lvis:
        def step():
            # a NEW batch of pixels every step (the clone is the "data loader"), then every chunk caption for it
            il = ImageList(imgs.clone(), [(H, W)] * Bn)
            if args.chunk_batch > 0:
                res = model.forward_chunks(il, chunks, max_items=args.chunk_batch)
                return res[-1]
            for cap, pm in chunks:
                out = model(il, captions=[cap] * Bn, positive_map=pm)
                if world > 1:
                    parallel.gather_detections(model.last_packed)
            return out
    else:
        og = parallel.OverlappedGather() if (world > 1 and args.overlap_gather) else None

        def step():
            out = model(images, captions=captions, positive_map=pmap)
            if og is not None:
                og.submit(model.last_packed)                        # the collective of this step runs under the next forward
            elif world > 1:
                parallel.gather_detections(model.last_packed)       # [world*B, 300, 6], one fixed-shape collective
            return out

    for _ in range(max(args.warmup, 2)):        # >= 2: first call autotunes eagerly, second captures the HIP graph
        step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if not lvis and og is not None:
        og.flush()                                                  # the last step's gather is inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    stats = dict(model.cache_stats)
    # per-kernel durations: HIP events around the launches (same kernels, same inputs) in a short EAGER pass --
    # inside a graph replay individual launches cannot be bracketed by events
    kern, prof_steps = {}, 0
    if rank == 0 and not lvis:
        prof_steps = min(3, args.steps)
        # single-stream for this pass: with the level / text streams active, concurrent kernels share the CUs and an
        # event-bracketed duration would include its neighbours (the timed region above keeps the multi-stream schedule)
        streams_on = cfg.MODEL.DYHEAD.LEVEL_STREAMS
        cfg.MODEL.DYHEAD.LEVEL_STREAMS = False
        try:
            ops.start_timing()
            for _ in range(prof_steps):
                step_out = model(images, captions=captions, positive_map=pmap)
            kern = ops.stop_timing()
            del step_out
        finally:
            cfg.MODEL.DYHEAD.LEVEL_STREAMS = streams_on

    if rank == 0:
        fwd_per_step = len(chunks) if lvis else 1
        ips = world * Bn * args.steps * fwd_per_step / dt
        roofs = kernel_rooflines(kern, max(prof_steps, 1), Bn, n_tok, embed=192 if large else 96) if kern else []
        hot = [r for r in roofs if r["bound"] == "mfma"]
        hot = [r for r in hot if not r.get("pair")]           # the dominant kernel = the SINGLE hand-written kernel with the most time per step
        roof = max(hot, key=lambda r: r["ms_per_step"]) if hot else None
        res = {
            "metric": "images/sec MQ-GLIP-T 800×1333 5-shot vision queries, 1/2/4/8 MI355X", "value": round(ips, 3), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[3]: MQ-GLIP-L (Swin-L window 12 + BERT-base + GCP + 8-layer VLDyHead), "
                                    f"{'bf16' if args.dtype == 'bf16' else 'fp16'} MFMA operands / fp32 accumulation, 5 vision queries x 40 classes, {n_tok}-token "
                                    "caption, every step a full forward") if large else
                                   ("BASELINE.json configs[1]: MQ-GLIP-T (Swin-T + BERT-base + GCP + 6-layer VLDyHead), 5 vision queries x "
                                    f"40 classes, {n_tok}-token caption, LVIS-style post-processing; every step a full forward (feature / "
                                    "language caches OFF; only host-side tokenizer / query-selection results are memoised)"
                                    + ("; SPLIT-PRECISE MODE: every MFMA contraction on fp32 operands split hi + lo into three fp16 MFMAs, fp32 library GEMMs"
                                       if args.dtype == "f32" else "") + ")") if not lvis else
                                   ("BASELINE.json configs[2] shape: MQ-GLIP-T, LVIS protocol -- 1203 synthetic categories in 31 chunk captions, "
                                    "each step = a new image batch x 31 forwards with the boundary's per-image feature cache and per-caption "
                                    "language cache ON; value counts FORWARDS (image x chunk) per second"
                                    + (f"; chunks stacked along the batch ({args.chunk_batch} items per launch sequence, model.forward_chunks)"
                                       if args.chunk_batch > 0 else "")),
                       "global_batch": world * Bn, "batch_per_gpu": Bn, "image": "800x1333 -> 800x1344",
                       "parallelism": f"dp{world}", "weights": "seeded random init (no checkpoints offline)",
                       "residual_streams": "fp32" if cfg.MODEL.get("RESIDUAL_FP32", True) else "fp16"},
            "detections_img0": len(out[0]),
            "comm": _comm_info(world),
            "hip_graph": bool(model.use_hip_graph and any(e.get("stage") == 2 for e in model._graphs.values())),
            "cache_stats": stats,
        }
        if lvis:
            res["forwards_per_step"] = fwd_per_step
            res["lvis_style_images_per_sec"] = round(ips / fwd_per_step, 3)
        else:
            gf = GFLOP_PER_IMAGE_L if large else GFLOP_PER_IMAGE
            gx = executed_gflop_per_image(n_tok, layers=8 if large else 6, swin=1625.0 if large else 198.2)
            res["model_tflops"] = round(ips * gx / 1e3, 2)
            res["model_frac_of_mfma_peak"] = round(ips * gx / 1e3 / (MFMA_PEAK_TFLOPS * world), 4)
            res["model_flops"] = {"executed_gflop_per_image": round(gx, 1), "reference_algorithmic_gflop_per_image": gf,
                                  "reference_algorithmic_tflops": round(ips * gf / 1e3, 2),
                                  "note": "model_tflops = EXECUTED flops (bench.executed_gflop_per_image: projections folded, K / V per unique vision "
                                          "token, live text blocks only) x images/s; the reference's un-folded count at T = 256 beside it"}
            res["roofline"] = roof
            res["rooflines"] = roofs
            res["kernels_ms_per_step"] = {k: round(v[1] / max(prof_steps, 1), 3) for k, v in sorted(kern.items())}
            res["kernel_selection"] = dict(ops.KERNELS)
            res["pmc_traffic_file"] = _pmc().get("_file")
            res["timing"] = "roofline records: HIP events on the launch stream around each launch, eager single-stream pass of the same steps"
            want_cpu = world == 1 and not args.no_cpu_baseline and not large
            if world == 1 and not args.no_lang_b64 and not large:
                try:
                    res["lang_path_b64"] = lang_path_b64(model, cfg, dev, chunks)
                except Exception as e:  # noqa: BLE001
                    res["lang_path_b64"] = {"error": repr(e)[:300]}
            if world == 1 and not args.no_experimental and not large and args.dtype == "f16" and not any(k in os.environ for k in tuple(ROUND3_KERNEL_SET) + tuple(ROUND4_KERNEL_SET) if k.startswith("MQ_")):
                # bounded subprocess lines, most informative first; each only while the whole run stays within a few minutes
                if time.perf_counter() - t_start < 200:
                    # the SPLIT-PRECISE mode (MODEL.COMPUTE_DTYPE = float32: fp32 operands carried as hi + lo through three fp16 MFMAs -- what the 1e-3
                    # parity tests run) on the SAME workload at the SAME batch: its images/s beside the fp16 line
                    res["split_precise"] = _sub_bench(["--dtype", "f32", "--batch", str(Bn), "--steps", "5", "--warmup", "2"], None, 150)
                    res["split_precise"].pop("workload", None)
                other = {}
                for key, argv, limit in (("configs[3] mq-glip-l bf16 B=4", ["--workload", "mq-glip-l", "--dtype", "bf16", "--steps", "5", "--warmup", "2"], 160),
                                         ("configs[4] mq-gdino-t B=16", ["--workload", "mq-gdino-t", "--steps", "5", "--warmup", "2"], 200),
                                         ("configs[2] lvis B=8 chunk-batch 32", ["--workload", "lvis", "--chunk-batch", "32", "--steps", "2", "--warmup", "2"], 240)):
                    if time.perf_counter() - t_start < limit:
                        other[key] = _sub_bench(argv, None, 150, keep=("roofline", "model_tflops", "lvis_style_images_per_sec", "forwards_per_step"))
                res["other_configs"] = other
            if want_cpu:
                # AFTER the GPU sub-runs, not beside them: GPU call 4 of round 6 ran the worker concurrently -- its median went from 5.7 to 12.2 s per
                # forward and the sub-runs lost 10 - 45 % (host threads of the replayed graphs compete with 32 pinned OpenMP threads)
                try:
                    res["cpu_baseline"] = cpu_baseline()
                except Exception as e:  # noqa: BLE001
                    res["cpu_baseline"] = {"error": repr(e)[:200]}
        print(compact_line(res, args.extras_file, write=bool(args.extras_file) or not args.no_extras), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


# ---- the ONE stdout line: contract fields only, < 4 KB; everything else goes to bench_extras.json ---------------------------
LINE_LIMIT = 4096
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
             "config", "roofline", "cpu_baseline", "comm", "hip_graph", "detections_img0", "model_tflops", "model_frac_of_mfma_peak",
             "forwards_per_step", "lvis_style_images_per_sec", "split_precise", "also_measured", "extras_file")
EXTRAS_FILE = "bench_extras.json"


def _short(v, n):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 1] + "~"


def compact_line(res, extras_path=None, write=True):
    """`res` (everything the run measured) -> the one JSON line of the contract.  The full record is written to bench_extras.json next to this
    script (and to gpurun_out/ when that directory exists, so that it comes back from a GPU box); the line keeps the contract fields, ONE roofline
    record, the CPU baseline and a few one-number summaries of the sub-runs, and names the extras file.  tests/test_host_cpu.py holds the size."""
    default_path = extras_path is None
    extras_path = extras_path or os.path.join(ROOT, EXTRAS_FILE)
    try:
        if not write:                                         # the bounded sub-runs of this script (--no-extras) leave the parent's file alone
            raise FileNotFoundError
        with open(extras_path, "w") as f:
            json.dump(res, f, indent=1)
        god = os.path.join(ROOT, "gpurun_out")
        if default_path and os.path.isdir(god):
            with open(os.path.join(god, EXTRAS_FILE), "w") as f:
                json.dump(res, f, indent=1)
    except FileNotFoundError:
        pass
    except OSError as e:                                      # a read-only checkout must not cost the line
        print(f"bench.py: could not write {extras_path}: {e}", file=sys.stderr)
    line = {k: res[k] for k in LINE_KEYS if k in res}
    if isinstance(line.get("config"), dict):
        line["config"] = {k: _short(v, 420) for k, v in line["config"].items()}
    roof = res.get("roofline")
    if isinstance(roof, dict):
        keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_step", "ms_per_step", "algorithmic_flops_per_launch",
                "algorithmic_bytes_per_launch", "l2_gather_GBs")
        line["roofline"] = {k: _short(roof[k], 160) for k in keep if k in roof}
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict):
        keep = ("value", "unit", "cores", "kind", "median_s_per_forward", "min_s_per_forward", "mq_glip_t", "sample", "error")
        line["cpu_baseline"] = {k: _short(cb[k], 260) for k in keep if k in cb}
    sp = res.get("split_precise")
    if isinstance(sp, dict):
        line["split_precise"] = {k: sp[k] for k in ("value", "unit", "ms_per_step", "batch_per_gpu", "dtype", "error") if k in sp}
        line["split_precise"]["note"] = "same workload and batch, MODEL.COMPUTE_DTYPE=float32: every MFMA on fp32 operands split hi+lo (1e-3 parity mode)"
    also = {}
    for key, sub in (res.get("other_configs") or {}).items():
        if isinstance(sub, dict):
            also[key] = sub.get("value", "error")
    ab = (res.get("kernel_set_ab") or {})
    for key, sub in ab.items():
        if isinstance(sub, dict) and "value" in sub:
            also[key] = sub["value"]
    lp = res.get("lang_path_b64")
    if isinstance(lp, dict) and "attention_mfma_utilisation" in lp:
        also["lang_path_b64"] = {"ms": lp.get("ms_language_path"), "attention_mfma_utilisation": lp.get("attention_mfma_utilisation"),
                                 "bert_fused_mfma_utilisation": (lp.get("bert_fused_launches") or {}).get("mfma_utilisation")}
    if also:
        line["also_measured"] = also
    if write:
        line["extras_file"] = os.path.basename(extras_path)
    out = json.dumps(line)
    for drop in ("also_measured", "comm", "model_frac_of_mfma_peak", "detections_img0"):        # never reached today; the line must parse whatever happens
        if len(out) < LINE_LIMIT:
            break
        line.pop(drop, None)
        out = json.dumps(line)
    if len(out) >= LINE_LIMIT:
        line["config"] = {"workload": _short(str((res.get("config") or {}).get("workload")), 200)}
        out = json.dumps(line)
    return out


if __name__ == "__main__":
    main()
