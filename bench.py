#!/usr/bin/env python
"""bench.py -- images/sec of the MQ-GLIP-T vision-language forward on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one `model(images, captions, positive_map)` forward (Swin -> FPN -> BERT+GCP -> VLDyHead -> ATSS
post-processing to list[BoxList]) on a batch of 8 synthetic 800x1333 images per GPU (BASELINE.json configs[1]:
MQ-GLIP-T, 5 vision queries per class, 40-class caption, fp16), inputs resident in HBM, followed (N > 1) by
the fixed-shape RCCL all-gather of the detections.  Weak scaling: per-GPU batch is fixed.
Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant hand-written kernel, timed
live with HIP events on the launch stream) and, at N = 1, `cpu_baseline` (the CPU oracle on a bounded sample).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 8
IMG_HW = (800, 1333)
NUM_CLASSES_IN_CAPTION = 40
# algorithmic work of one image-forward (2*MAC), BASELINE.md section 2 / SURVEY.md 8(d)
GFLOP_PER_IMAGE = 1451.0
MFMA_PEAK_TFLOPS = 2500.0          # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def build_model(dev, full=True):
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
    if os.environ.get("MQ_FUSED_DCN") is not None:     # A/B switch between the two DCNv2 implementations (same results)
        cfg.MODEL.DYHEAD.FUSED_DCN = os.environ["MQ_FUSED_DCN"] == "1"
    tok_dir = build_synthetic_tokenizer(tempfile.mkdtemp(prefix="mqdet_tok_"))
    cfg.MODEL.LANGUAGE_BACKBONE.TOKENIZER_TYPE = tok_dir
    tk = AutoTokenizer.from_pretrained(tok_dir)
    model = GeneralizedVLRCNN_New(cfg, tokenizer=tk)
    randomize_(model, seed=0)
    caption, spans = synthetic_caption(NUM_CLASSES_IN_CAPTION)
    pmap = positive_map_from_spans(tk, caption, spans, list(range(1, NUM_CLASSES_IN_CAPTION + 1)))
    model.load_query_bank(synthetic_bank(pmap.keys(), cfg.MODEL.BACKBONE.OUT_CHANNELS, cfg.VISION_QUERY.NUM_QUERY_PER_CLASS))
    model.to(dev)
    model.prepare(dev)
    return cfg, model, caption, pmap


CPU_BASELINE_THREADS = 32      # the oracle's many small torch ops stop scaling (and can crawl) far below 256 threads


def _cpu_baseline_worker():
    """CPU oracle (pure-PyTorch fp32 restatement of the reference; the reference itself has no CPU path) on
    one 800x1333 image, a few forwards (bounded sample)."""
    from oracle import glip_t_spec
    from oracle import detector as od
    from oracle.weights import make_state_dict, make_query_bank
    threads = min(os.cpu_count() or 1, CPU_BASELINE_THREADS)
    torch.set_num_threads(threads)
    spec = glip_t_spec()
    sd = make_state_dict(spec, 0)
    g = torch.Generator().manual_seed(0)
    images, sizes = od.pad_images([torch.randn(3, *IMG_HW, generator=g)], 32)
    T = spec.max_query_len
    nvalid = 2 * NUM_CLASSES_IN_CAPTION + 1
    ids = torch.zeros(1, T, dtype=torch.long)
    ids[:, :nvalid] = torch.randint(1000, spec.vocab, (nvalid,), generator=g)
    am = torch.zeros(1, T, dtype=torch.long)
    am[:, :nvalid] = 1
    pm = {i + 1: [1 + 2 * i] for i in range(NUM_CLASSES_IN_CAPTION)}
    bank = make_query_bank(pm.keys(), spec)
    n_fwd = 3                      # ~20 s of CPU work on the GPU box's host
    t = time.time()
    for _ in range(n_fwd):
        od.forward(sd, spec, images, sizes, ids, am, pm, bank)
    dt = time.time() - t
    print(json.dumps({"value": round(n_fwd / dt, 4), "unit": "images/sec", "cores": threads, "kind": "port",
                      "sample": f"{n_fwd} forwards of the fp32 CPU oracle on one 800x1333 image (padded 800x1344), {dt:.1f} s, "
                                f"{threads} torch threads on a {os.cpu_count()}-core host"}), flush=True)


def cpu_baseline(timeout=300):
    """Run the worker in a subprocess with a hard time limit so the default bench run stays bounded."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(CPU_BASELINE_THREADS), MKL_NUM_THREADS=str(CPU_BASELINE_THREADS),
               HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], env=env,
                           capture_output=True, text=True, timeout=timeout)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": (r.stderr or r.stdout)[-300:]}
    except subprocess.TimeoutExpired:
        return {"error": f"cpu baseline worker exceeded {timeout} s", "cores": CPU_BASELINE_THREADS, "kind": "port"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=B_PER_GPU)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay (for PMC profiling)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return _cpu_baseline_worker()

    from mq_det_amd import parallel
    from mq_det_amd import ops
    from mq_det_amd.structures import ImageList
    rank, local, world = parallel.init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ops.load_library()
    cfg, model, caption, pmap = build_model(dev)
    if args.no_graph:
        model.use_hip_graph = False

    Bn = args.batch
    g = torch.Generator().manual_seed(1000 + rank)
    H, W = IMG_HW
    Hp, Wp = -(-H // 32) * 32, -(-W // 32) * 32
    imgs = torch.zeros(Bn, 3, Hp, Wp)
    imgs[:, :, :H, :W] = torch.randn(Bn, 3, H, W, generator=g)
    images = ImageList(imgs.to(dev), [(H, W)] * Bn)
    captions = [caption] * Bn
    K = cfg.MODEL.ATSS.DETECTIONS_PER_IMG

    def step():
        out = model(images, captions=captions, positive_map=pmap)
        if world > 1:
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
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    # per-kernel durations: HIP events around the launches (same kernels, same inputs) in a short EAGER pass --
    # inside a graph replay individual launches cannot be bracketed by events
    kern, prof_steps = {}, 0
    if rank == 0:
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
        ips = world * Bn * args.steps / dt
        # dominant hand-written kernels: the two VLFuse attention kernels (image side / text side, 6 + 6 launches / forward;
        # the text-side time includes its split-merge launch).
        # Algorithmic FLOPs per launch = QK^T + PV = 4 * B * heads * Nq * Nk * 256 (2*MAC each), DESIGN.md section 3.
        # Image->text launches only visit the key tiles that hold real caption tokens (padding is masked to an exact
        # zero contribution and skipped), so their work is counted with the visited keys, not with T = 256.
        N_img = sum(h * w for h, w in [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)])
        n_tok = int(model.tokenize(captions, dev)[1][0].sum())
        nk_vis = min(256, -(-n_tok // 64) * 64)
        # Text->image launches only compute the 128-row query tiles that hold real caption tokens (all-padding tiles are
        # skipped and come back as zeros), so their work is counted with those rows only.
        tq_live = min(256, -(-n_tok // 128) * 128)
        fl = {"i2t": 4.0 * Bn * 8 * N_img * nk_vis * 256, "t2i": 4.0 * Bn * 8 * tq_live * N_img * 256}
        roof = None
        i2t = [v for k, v in kern.items() if k.startswith("vlfuse_i2t_n%d_" % N_img)]
        t2i = [v for k, v in kern.items() if k.startswith("vlfuse_t2i_n%d_" % N_img)]
        if i2t and t2i:
            n_l = i2t[0][0] + t2i[0][0]
            ms = i2t[0][1] + t2i[0][1]
            flops = i2t[0][0] * fl["i2t"] + t2i[0][0] * fl["t2i"]
            ach = flops / (ms * 1e-3) / 1e12
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_vlfuse.json")
            if os.path.exists(pmc):                 # PMC passes are separate rocprofv3 runs (see profiles/README.md)
                traffic = json.load(open(pmc)).get("traffic_bytes_per_launch_avg")
            roof = {"bound": "mfma", "kernel": "vlfuse_i2t_kernel + vlfuse_t2i_kernel (VLFuse image<->text attention, 8 heads x 256)",
                    "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                    "traffic": traffic, "avg_launch_ms": round(ms / n_l, 4), "launches": n_l,
                    "flops_per_launch": {"image_to_text": fl["i2t"], "text_to_image": fl["t2i"], "visited_text_keys": nk_vis, "computed_text_rows": tq_live},
                    "per_direction_tflops": {"image_to_text": round(fl["i2t"] * i2t[0][0] / (i2t[0][1] * 1e-3) / 1e12, 1),
                                             "text_to_image": round(fl["t2i"] * t2i[0][0] / (t2i[0][1] * 1e-3) / 1e12, 1)},
                    "timing": "HIP events on the launch stream around each launch, eager pass of the same steps"}
        res = {
            "metric": "images/sec MQ-GLIP-T 800\u00d71333 5-shot vision queries, 1/2/4/8 MI355X", "value": round(ips, 3), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: MQ-GLIP-T (Swin-T + BERT-base + GCP + 6-layer VLDyHead), "
                                   "5 vision queries x 40 classes, caption padded to 256 tokens, LVIS-style post-processing",
                       "global_batch": world * Bn, "batch_per_gpu": Bn, "image": "800x1333 -> 800x1344",
                       "parallelism": f"dp{world}", "weights": "seeded random init (no checkpoints offline)"},
            "model_tflops": round(ips * GFLOP_PER_IMAGE / 1e3, 2),
            "model_frac_of_mfma_peak": round(ips * GFLOP_PER_IMAGE / 1e3 / (MFMA_PEAK_TFLOPS * world), 4),
            "detections_img0": len(out[0]),
            "roofline": roof,
            "kernels_ms_per_step": {k: round(v[1] / max(prof_steps, 1), 3) for k, v in sorted(kern.items())},
            "hip_graph": bool(model.use_hip_graph and any(e.get("stage") == 2 for e in model._graphs.values())),
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # noqa: BLE001
                res["cpu_baseline"] = {"error": repr(e)[:200]}
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
