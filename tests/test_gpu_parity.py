"""GPU parity tests proper: the HIP path (through the C ABI of libmqdet_hip.so) vs the CPU oracle on the
same seeded inputs.  Tolerances are stated in tests/parity_checks.py."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    from mq_det_amd import ops
    ops.load_library()            # fails loudly when the in-tree .so is missing
    return torch.device("cuda:0")


def _assert(res):
    res = res if isinstance(res, list) else [res]
    out = os.environ.get("MQ_LADDER_OUT")                # optional: every row of every check as a JSON line (profiles/*_ladder.jsonl)
    if out:
        import json
        with open(out, "a") as f:
            for r in res:
                f.write(json.dumps({k: v for k, v in r.items() if isinstance(v, (int, float, str, bool))}) + "\n")
    bad = [r for r in res if not r["ok"]]
    assert not bad, "\n".join(f"{r['name']}: max_err={r['max_err']:.3e} norm={r['norm_err']:.3e} mean={r['mean_err']:.3e} tol={r['tol']}"
                              + (f" | x floor: mean {r['ratio_mean']:.2f} max {r['ratio_max']:.2f} ({r['gate']})" if "ratio_mean" in r else "")
                              for r in bad)


def test_vlfuse_strided_operands_equal_contiguous(dev):
    """ABI 30: the VLFuse kernels read the folded keys / values through element strides.  The views the fusion layer passes (slices of ONE
    projection output [B, T, heads*256 | heads*256 | pad]) must give bit for bit what contiguous [B, heads, T, 256] copies give -- same
    arithmetic, other addresses -- at the benchmark geometry's caption length and at a short one."""
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(30)
    for B, Hh, N, T, kv in ((2, 8, 2000, 144, (141, 97)), (1, 4, 777, 48, None)):
        pr = (torch.randn(B, T, 2 * Hh * 256 + 16, generator=g) / 8).half().to(dev)
        kf_v = pr[..., :Hh * 256].unflatten(-1, (Hh, 256)).permute(0, 2, 1, 3)
        vo_v = pr[..., Hh * 256:2 * Hh * 256].unflatten(-1, (Hh, 256)).permute(0, 2, 1, 3)
        assert not kf_v.is_contiguous()
        kf_c, vo_c = kf_v.contiguous(), vo_v.contiguous()
        v_ln = torch.randn(B, N, 256, generator=g).half().to(dev)
        bias = torch.randn(B, Hh, T, generator=g).to(dev)
        ob = torch.randn(256, generator=g).half().to(dev)
        kv_len = None if kv is None else torch.tensor(kv, dtype=torch.int32, device=dev)
        mk = 0 if kv is None else max(kv)
        a = ops.vlfuse_i2t(v_ln, kf_v, vo_v, bias, ob, kv_len=kv_len, max_kv=mk)
        b = ops.vlfuse_i2t(v_ln, kf_c, vo_c, bias, ob, kv_len=kv_len, max_kv=mk)
        assert torch.equal(a, b), float((a.float() - b.float()).abs().max())
        a = ops.vlfuse_t2i(kf_v, v_ln, 3, kv_len=kv_len, max_kv=mk)
        b = ops.vlfuse_t2i(kf_c, v_ln, 3, kv_len=kv_len, max_kv=mk)
        assert torch.equal(a, b), float((a.float() - b.float()).abs().max())


def test_pooled_tokens_fused_equals_torch(dev):
    """mq_pool2x2_tokens_fwd against the torch statement it replaces (five F.avg_pool2d + concat, generalized_vl_rcnn_new.py:291-293) on the device,
    at the benchmark pyramid and at odd sizes: bit for bit."""
    import torch.nn.functional as F
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(31)
    for dt in (torch.float16, torch.bfloat16):
        for B, sizes in ((8, ((100, 168), (50, 84), (25, 42), (13, 21), (7, 11))), (2, ((9, 7), (5, 4), (3, 2)))):
            feats = [torch.randn(B, h, w, 256, generator=g).to(dt).to(dev).permute(0, 3, 1, 2) for h, w in sizes]
            ref = torch.cat([F.avg_pool2d(f, 2).permute(0, 2, 3, 1).flatten(1, 2) for f in feats], 1)
            got = ops.pool2x2_tokens(feats)
            assert got.shape == ref.shape and torch.equal(got, ref), (dt, sizes, float((got.float() - ref.float()).abs().max()))


ATTN = [
    dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True),
    dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True, clamp=50000.0, big=True),
    dict(B=1, H=8, D=32, Nq=200, Nk=5577, nsplit=4),
    dict(B=1, H=8, D=32, Nq=37, Nk=61),
    dict(B=2, H=8, D=32, Nq=1, Nk=9),
    dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True, kvlen=True),
]


@pytest.mark.parametrize("cfg", ATTN, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_attention(dev, cfg):
    import parity_checks as pc
    _assert(pc.check_attention(dev, **cfg))


def test_attention_strided_views(dev):
    import parity_checks as pc
    _assert(pc.check_attention_strided(dev))


@pytest.mark.parametrize("name", ["check_window_attention", "check_swin_fpn", "check_gcp_block", "check_pre_select",
                                  "check_vlfuse_kernels", "check_vl_fuse", "check_dcn", "check_conv3x3", "check_layernorm", "check_dyconv", "check_nms", "check_full_model",
                                  "check_ref_pins", "check_post_golden", "check_roi_align", "check_extract_query", "check_swin_mlp", "check_msdeform_attn",
                                  "check_align_fused", "check_fusion_layer", "check_post_fused", "check_attention_text", "check_bert_attn_qkv", "check_gcp_attn_fused", "check_patch_embed",
                                  "check_bert_clamp_fused"])
def test_block(dev, name):
    import parity_checks as pc
    _assert(getattr(pc, name)(dev))


def test_full_model_without_vision_queries(dev):
    """Plain GLIP-T path (no query bank), B = 1 -- BASELINE.json configs[0] shape."""
    import parity_checks as pc
    _assert(pc.check_full_model(dev, vision_queries=False))


@pytest.mark.parametrize("clamp", [False, True])
def test_bert_layer(dev, clamp):
    import parity_checks as pc
    _assert(pc.check_bert_layer(dev, clamp))


def test_boundary_returns_boxlists(dev):
    """model(images, captions=..., positive_map=...) -> list[BoxList] with scores/labels (boundary contract)."""
    import tempfile
    import parity_checks as pc
    from transformers import AutoTokenizer
    from mq_det_amd import BoxList
    from mq_det_amd.structures import to_image_list
    from mq_det_amd.utils.tokenizer import build_synthetic_tokenizer, synthetic_caption, positive_map_from_spans
    spec, sd, cfg, model, P = pc.tiny(dev)
    tk = AutoTokenizer.from_pretrained(build_synthetic_tokenizer(tempfile.mkdtemp(), size=spec.vocab))
    model.tokenizer = tk
    cap, spans = synthetic_caption(6)
    pm = positive_map_from_spans(tk, cap, spans, list(range(1, 7)))
    from oracle.weights import make_query_bank
    model.load_query_bank(make_query_bank(pm.keys(), spec))
    imgs = to_image_list([torch.randn(3, 150, 190), torch.randn(3, 160, 170)], 32).to(dev)
    out = model(imgs, captions=[cap, cap], positive_map=pm)
    assert len(out) == 2 and all(isinstance(o, BoxList) for o in out)
    for o, (h, w) in zip(out, imgs.image_sizes):
        assert o.mode == "xyxy" and o.size == (w, h)
        assert o.get_field("scores").dtype == torch.float32 and o.get_field("labels").dtype == torch.int64
        assert len(o) <= cfg.MODEL.ATSS.DETECTIONS_PER_IMG
        if len(o):
            assert o.bbox[:, 0].min() >= 0 and o.bbox[:, 2].max() <= w - 1 and o.bbox[:, 3].max() <= h - 1
            assert set(o.get_field("labels").tolist()) <= set(pm.keys())


def _equal_detections(a, b):
    """BIT-EQUAL detections.  `tools/replay_equality_probe.py` on the MI355X (GPU call 4 of round 6, profiles/r06_call4_replay_equality.txt):
    an eager forward, the capture and every replay of its HIP graph produce identical bytes under every BLAS setting tried -- the library GEMMs pick the
    same kernel for the same shape in and out of capture, and every hand-written kernel is bitwise reproducible (tests/determinism_diag.py) -- so
    the same program on the same inputs is compared with torch.equal, not matched."""
    return (len(a) == len(b) and torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("scores"), b.get_field("scores"))
            and torch.equal(a.get_field("labels"), b.get_field("labels")))


def _same_detections(a, b, frac=0.9, dscore=0.03, iou_min=0.9):
    """Order-insensitive IoU matching, ONLY for comparisons ACROSS batch shapes (chunk batching, micro-batch lanes): the library GEMMs pick their
    kernel by problem size, so the fp32 summation order differs and -- on the tiny random-init model of these tests, whose scores crowd the
    threshold -- near-threshold detections come and go (GPU call 5 of round 6 tried frac 0.99 / |d score| 1e-3 here: 118 vs 119 boxes fails it).
    The SAME program on the same inputs is compared with _equal_detections."""
    if len(b) == 0:
        return len(a) == 0
    ab, asc, al = a.bbox.cpu(), a.get_field("scores").cpu(), a.get_field("labels").cpu()
    hit = 0
    for box, sc, lab in zip(b.bbox.cpu(), b.get_field("scores").cpu(), b.get_field("labels").cpu()):
        lt, rb = torch.max(ab[:, :2], box[:2]), torch.min(ab[:, 2:], box[2:])
        inter = (rb - lt + 1).clamp(min=0).prod(1)
        iou = inter / ((ab[:, 2] - ab[:, 0] + 1) * (ab[:, 3] - ab[:, 1] + 1) + (box[2] - box[0] + 1) * (box[3] - box[1] + 1) - inter)
        ok = (al == lab) & (iou > iou_min) & ((asc - sc).abs() < dscore)
        hit += bool(ok.any())
    return abs(len(a) - len(b)) <= max(3, len(b) // 20) and hit >= frac * len(b)


def test_hip_graph_replay_matches_eager(dev):
    """Third call of the same shapes replays the captured HIP graph; detections must agree with the eager ones."""
    import parity_checks as pc
    from mq_det_amd.structures import ImageList
    spec, sd, cfg, model, P = pc.tiny(dev)
    images, sizes, ids, am, pm, bank = pc.make_inputs(spec)
    model.load_query_bank(bank)
    il = ImageList(images.to(dev), sizes)
    kw = dict(captions=None, positive_map=pm, input_ids=ids.to(dev), attention_mask=am.to(dev))
    model.use_hip_graph = False
    ref = model(il, **kw)
    model.use_hip_graph = True
    model.clear_caches()
    outs = [model(il, **kw) for _ in range(4)]           # eager, capture+replay, replay, replay
    assert any(e.get("stage") == 2 for e in model._graphs.values()), "HIP graph was not captured"
    for out in outs:                                     # eager warm-up, capture + first replay, replays: the same bytes as the eager forward
        for a, b in zip(out, ref):
            assert _equal_detections(a, b)
    for a, b in zip(outs[2], outs[3]):                   # two replays of one graph
        assert _equal_detections(a, b)
    # new pixels through the same graph
    il2 = ImageList(torch.flip(images, dims=[3]).to(dev), sizes)
    model.use_hip_graph = False
    ref2 = model(il2, **kw)
    model.use_hip_graph = True
    out2 = model(il2, **kw)
    for a, b in zip(out2, ref2):
        assert _equal_detections(a, b)


def test_staggered_micro_batches_match_the_single_lane_forward(dev):
    """MODEL.MICRO_BATCHES = 2 (the batch as two lanes staggered one stage apart, detector._staggered_program): same detections per image as
    the single-lane forward, eagerly and from the replayed HIP graph; odd batch sizes split 2 + 1."""
    import parity_checks as pc
    from mq_det_amd.structures import ImageList
    spec, sd, cfg, model, P = pc.tiny(dev)
    images, sizes, ids, am, pm, bank = pc.make_inputs(spec)
    model.load_query_bank(bank)
    prev_mb, prev_cache, prev_graph = model.micro_batches, model.backbone_cache, model.use_hip_graph
    try:
        for reps in (2, 3):                                  # B = 4 (2 + 2) and B = 6 ... (tiny inputs hold 2 images)
            imgs = torch.cat([images] + [torch.flip(images, dims=[3 - (r % 2)]) * (1.0 - 0.1 * r) for r in range(1, reps)])
            il = ImageList(imgs.to(dev), list(sizes) * reps)
            kw = dict(captions=None, positive_map=pm, input_ids=ids.repeat(reps, 1).to(dev), attention_mask=am.repeat(reps, 1).to(dev))
            model.backbone_cache, model.use_hip_graph, model.micro_batches = False, False, 1
            model.clear_caches()
            ref = model(il, **kw)
            model.micro_batches = 2 if reps == 2 else 4       # 4 lanes over 6 images: 2 + 2 + 2 (per = ceil(6 / 4) = 2)
            lanes = model(il, **kw)
            assert model.cache_stats["eager"] > 0
            for a, b in zip(lanes, ref):
                assert _same_detections(a, b, frac=0.95)
            model.use_hip_graph = True
            model.clear_caches()
            outs = [model(il, **kw) for _ in range(3)]
            assert any(k[0] == "_staggered_program" and e.get("stage") == 2 for k, e in model._graphs.items()), "staggered program was not captured"
            for a, b in zip(outs[2], ref):
                assert _same_detections(a, b, frac=0.95)
            for a, b in zip(outs[2], outs[1]):                # capture + replay vs replay of the staggered program: the same bytes
                assert _equal_detections(a, b)
    finally:
        model.micro_batches, model.backbone_cache, model.use_hip_graph = prev_mb, prev_cache, prev_graph
        model.clear_caches()


def test_hip_graph_capture_with_process_group(dev):
    """bench.py --gpus N initialises torch.distributed before the first forward: the HIP-graph capture must survive the
    RCCL watchdog thread (capture_error_mode="thread_local"); world size 1 here, one GPU."""
    import os
    import socket
    import torch.distributed as dist
    import parity_checks as pc
    from mq_det_amd import parallel
    from mq_det_amd.structures import ImageList
    created = False
    if not dist.is_initialized():
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        spec, sd, cfg, model, P = pc.tiny(dev)
        images, sizes, ids, am, pm, bank = pc.make_inputs(spec)
        model.load_query_bank(bank)
        il = ImageList(images.to(dev), sizes)
        kw = dict(captions=None, positive_map=pm, input_ids=ids.to(dev), attention_mask=am.to(dev))
        model.use_hip_graph = True
        model.clear_caches()
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)                                   # make sure the communicator (and its watchdog) is alive
        outs = [model(il, **kw) for _ in range(3)]
        assert any(e.get("stage") == 2 for e in model._graphs.values()), "HIP graph was not captured under a process group"
        g = parallel.gather_detections(model.last_packed)
        assert g.shape == model.last_packed.shape
        for a, b in zip(outs[1], outs[2]):
            assert _equal_detections(a, b)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_detection_and_evaluator_gather_over_rccl(dev, world):
    """SURVEY 8(e) / 8(f-4) on the device: the fixed-shape all-gather of detections (also one step behind: OverlappedGather) and
    the evaluator's top-k exchange run over RCCL with their state in HBM -- one rank here (the collectives are forced), and two ranks on the
    first box that has two GPUs (skipped below that), one process per GPU under torch.distributed.run like bench.py --gpus N."""
    import socket
    import subprocess
    import sys
    if torch.cuda.device_count() < world:
        pytest.skip(f"{world} GPUs needed, {torch.cuda.device_count()} visible")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = str(s_.getsockname()[1])
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", port, script], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("RCCL_GATHER_OK") == world


@pytest.mark.parametrize("caption,hw", [("short", ((800, 1333), (736, 1280))), ("long", ((800, 1333),))],
                         ids=["81-token-caption-B2", "141-token-caption-B1"])
def test_benchmark_configuration_parity(dev, caption, hw):
    """Full-depth MQ-GLIP-T (the configuration bench.py times) on 800x1333 images vs the fp32 oracle: per-stage error ladder
    (Swin -> FPN -> language backbone -> each of the 6 fusion layers -> heads), class scores, and >= 95 % of the oracle's
    top-100 detections reproduced in both score-aggregation modes.  THE GATE: every stage's error <= a stated multiple of the
    committed fp16-operand floor of that stage (tests/golden/floor_bench.json; parity_checks.FLOOR_RATIO_*)."""
    import parity_checks as pc
    _assert(pc.check_benchmark_config(dev, caption, hw))


def test_benchmark_configuration_b8_graph_replay(dev):
    """The configuration bench.py TIMES -- B = 8 images 800x1333, 141-token caption, the forward replayed from a HIP graph -- against the
    oracle: stage rows of batch item 0 under the floor gate, replayed detections vs the oracle / the eager forward / B = 1 forwards."""
    import parity_checks as pc
    _assert(pc.check_benchmark_b8_graph(dev))


def test_mq_glip_l_benchmark_configuration_parity_fp16(dev):
    """BASELINE configs[3] at FULL depth: MQ-GLIP-L (Swin-L 2-2-18-2, window 12, 8 fusion layers) on one 800x1333 image, 141-token
    caption, fp16 operands, against the fp32 oracle -- same floor gate (configs/pretrain/mq-glip-l.yaml:11-41)."""
    import parity_checks as pc
    _assert(pc.check_benchmark_config(dev, "long", ((800, 1333),), family="l"))


def test_backbone_and_caption_caches(dev):
    """same pixels again -> cached Swin / FPN features ("rest" program), same caption again -> cached
    image-independent BERT layers; results identical to the uncached forward; an in-place change of the pixels or a new
    tensor is a miss; the HIP-graph cache is bounded (LRU) and keyed by shapes only."""
    import tempfile
    import parity_checks as pc
    from transformers import AutoTokenizer
    from mq_det_amd.structures import ImageList
    from mq_det_amd.utils.tokenizer import build_synthetic_tokenizer, synthetic_caption, positive_map_from_spans
    from oracle.weights import make_query_bank
    spec, sd, cfg, model, P = pc.tiny(dev)
    tk = AutoTokenizer.from_pretrained(build_synthetic_tokenizer(tempfile.mkdtemp(), size=spec.vocab))
    model.tokenizer = tk
    caps = []
    for c in range(3):
        cap, spans = synthetic_caption(6, start=10 * c, words=(1, 2))
        caps.append((cap, positive_map_from_spans(tk, cap, spans, list(range(1 + 6 * c, 7 + 6 * c)))))
    model.load_query_bank(make_query_bank(range(1, 19), spec))
    images, sizes, *_ = pc.make_inputs(spec)
    model.clear_caches()
    model.backbone_cache, model.use_hip_graph = False, False
    il = ImageList(images.to(dev), sizes)
    ref = [model(il, captions=[cap] * 2, positive_map=pm) for cap, pm in caps]
    model.backbone_cache, model.use_hip_graph = True, True
    model.cache_stats = {k: 0 for k in model.cache_stats}
    for rep in range(4):                                    # the LVIS protocol: every caption for the same pixels, image after image
        il = ImageList(images.to(dev).clone(), sizes)
        for (cap, pm), r in zip(caps, ref):
            out = model(il, captions=[cap] * 2, positive_map=pm)
            for a, b in zip(out, r):
                # a cache hit runs the SAME kernels on the same values (the cached features ARE the uncached forward's): the same bytes.  (The
                # image-independent BERT layers of a cached caption were computed for ONE caption row set and repeated: same per-row arithmetic.)
                assert _equal_detections(a, b), (rep, cap[:20])
    st = model.cache_stats
    assert st["backbone_miss"] == 4 and st["backbone_hit"] == 8, st
    assert st["front_hit"] >= 6 and st["graph_replay"] >= 4, st
    # chunk batching: every caption for the same pixels in ONE call (chunks stacked along the batch dimension)
    il = ImageList(images.to(dev).clone(), sizes)
    for _ in range(3):                                      # eager, capture, replay
        batched = model.forward_chunks(il, caps)
    assert len(batched) == len(caps)
    for out, r in zip(batched, ref):
        for a, b in zip(out, r):
            assert _same_detections(a, b)
    il.tensors.add_(0.25)                                   # in-place change of the cached pixels -> version bump -> miss
    miss = model.cache_stats["backbone_miss"]
    model(il, captions=[caps[0][0]] * 2, positive_map=caps[0][1])
    assert model.cache_stats["backbone_miss"] == miss + 1
    # image sizes are a tensor input, not a graph key: other (h, w) of the same padded shape replay the same graph
    n_graphs = len(model._graphs)
    il2 = ImageList(images.to(dev).clone(), [(h - 3, w - 5) for (h, w) in sizes])
    model(il2, captions=[caps[0][0]] * 2, positive_map=caps[0][1])
    assert len(model._graphs) == n_graphs
    model.graph_cache_size = 2
    for n_img in (1, 2, 1):                                 # new shapes -> new keys; LRU evicts beyond 2 graphs
        il3 = ImageList(images[:n_img].to(dev).clone(), sizes[:n_img])
        for _ in range(3):
            model(il3, captions=[caps[1][0]] * n_img, positive_map=caps[1][1])
    assert len(model._graphs) <= 2 and model.cache_stats["graph_evict"] >= 1
    model.graph_cache_size = 8
    model.clear_caches()


def test_mq_glip_l_family(dev):
    """BASELINE configs[3] model family at tiny depth: Swin-L widths / heads, window 12 (144-token windows through the
    160-padded window-attention kernel), FPN on 384 / 768 / 1536 channels, the full MQ-GLIP forward behind it."""
    import parity_checks as pc
    _assert(pc.check_window_attention(dev, large=True))
    _assert(pc.check_swin_fpn(dev, large=True))
    _assert(pc.check_full_model(dev, large=True))


# ------------------------------------------------------------------------------------------------ MQ-GroundingDINO (configs[4])
@pytest.mark.parametrize("name", ["check_attention_qk_mask", "check_vlfuse_heads_mask", "check_msdeform_attn_q", "check_gdino_tiny",
                                  "check_gdino_state_dict_and_quirks"])
def test_groundingdino_block(dev, name):
    import gdino_checks as gc
    _assert(getattr(gc, name)(dev))


def test_groundingdino_benchmark_config(dev):
    """Full-depth MQ-GroundingDINO-T on one 800 x 1333 image, 40 classes x 5 vision queries, vs the fp32 oracle."""
    import gdino_checks as gc
    _assert(gc.check_gdino_benchmark_config(dev))


# ------------------------------------------------------------------------------------------------ INTEGRATION.md stubs
def _integration_stubs():
    """The ctypes stubs of INTEGRATION.md section 2, executed as written (only the library path is filled in)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", text, re.S) if "_lib" in b]
    code = "\n".join(blocks).replace("/path/to/mq_det_amd/lib/libmqdet_hip.so", os.path.join(root, "mq_det_amd", "lib", "libmqdet_hip.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    return ns


def test_integration_md_operator_stubs_run_as_written(dev):
    """VERDICT r1 (row b): the reference-side binding shown to a maintainer is executed, not only printed: `ml_nms`, `nms`,
    `modulated_deform_conv_forward`, `roi_align_forward`, `ms_deform_attn_forward` from INTEGRATION.md against the oracle."""
    from oracle import gdino as og, head as ohead, postprocess as opost, roi as oroi
    ns = _integration_stubs()
    g = torch.Generator().manual_seed(91)
    # _C.ml_nms
    n = 300
    xy = torch.rand(n, 2, generator=g) * 200
    boxes = torch.cat([xy, xy + 20 + torch.rand(n, 2, generator=g) * 60], 1)
    scores, labels = torch.rand(n, generator=g), torch.randint(1, 4, (n,), generator=g)
    keep = ns["ml_nms"](boxes.to(dev), scores.to(dev), labels.to(dev), 0.6).cpu()
    assert torch.equal(keep, opost.ml_nms(boxes, scores, labels, 0.6))
    # _C.nms (class-agnostic; layers/nms.py:2-8 falls back to it without torchvision): one label for all boxes
    keep = ns["nms"](boxes.to(dev), scores.to(dev), 0.5).cpu()
    assert torch.equal(keep, opost.ml_nms(boxes, scores, torch.zeros(n, dtype=torch.long), 0.5)) and 0 < len(keep) < n
    assert ns["nms"](boxes[:0].to(dev), scores[:0].to(dev), 0.5).shape == (0,)
    # _C.modulated_deform_conv_forward
    x = torch.randn(2, 256, 20, 24, generator=g).half().float()
    off = torch.randn(2, 18, 20, 24, generator=g) * 1.5
    mask = torch.sigmoid(torch.randn(2, 9, 20, 24, generator=g))
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).half().float()
    b = (torch.randn(256, generator=g) * 0.1).half().float()
    mask[0, 3, 5, 6], mask[1, 0, 0, 0] = 0.0, 1.0                # probabilities AT the ends: a logit round trip would give +-inf
    for stride in (1, 2):                                        # the reference's own call, layers/deform_conv.py:184-204 (19 arguments)
        Ho, Wo = (20 - 1) // stride + 1, (24 - 1) // stride + 1
        output = x.new_empty(2, 256, Ho, Wo).to(dev)
        ones, columns = torch.empty(0, device=dev), torch.empty(0, device=dev)
        ret = ns["modulated_deform_conv_forward"](x.to(dev), w.to(dev), b.to(dev), ones, off[:, :, :Ho, :Wo].contiguous().to(dev),
                                                  mask[:, :, :Ho, :Wo].contiguous().to(dev), output, columns, 3, 3, stride, stride, 1, 1, 1, 1, 1, 1, True)
        assert ret is output
        ref = ohead.dcn_v2(x, off[:, :, :Ho, :Wo].contiguous(), mask[:, :, :Ho, :Wo].contiguous(), w, b, stride)
        assert float((output.float().cpu() - ref).abs().max()) <= 4e-3 * max(1.0, float(ref.abs().max()))
    # _C.roi_align_forward (ROIAlignV2)
    feat = torch.randn(2, 64, 30, 40, generator=g)
    rois = torch.tensor([[0, 3.0, 4.0, 30.0, 25.0], [1, 10.5, 2.25, 39.0, 29.0], [0, 0.0, 0.0, 8.0, 8.0]])
    out = ns["roi_align_forward"](feat.to(dev), rois.to(dev), 0.25, 7, 7, 2, aligned=True).cpu()
    torch.testing.assert_close(out, oroi.roi_align(feat, rois, 7, 0.25, 2, aligned=True), atol=1e-4, rtol=1e-4)
    # groundingdino_new._C.ms_deform_attn_forward
    shapes = [(12, 16), (6, 8), (3, 4), (2, 2)]
    S = sum(h * w for h, w in shapes)
    value = torch.randn(2, S, 8, 32, generator=g)
    loc = torch.rand(2, 50, 8, 4, 4, 2, generator=g) * 1.1 - 0.05
    attn = torch.rand(2, 50, 8, 16, generator=g).softmax(-1).reshape(2, 50, 8, 4, 4)
    hw = torch.tensor(shapes)
    start = torch.cat([hw.new_zeros(1), (hw[:, 0] * hw[:, 1]).cumsum(0)[:-1]])
    out = ns["ms_deform_attn_forward"](value.to(dev), hw.to(dev), start.to(dev), loc.to(dev), attn.to(dev)).cpu()
    torch.testing.assert_close(out, og.ms_deform_attn_core(value, shapes, loc, attn), atol=1e-4, rtol=1e-4)


# ------------------------------------------------------------------------------------------------ added after the last GPU call of round 2
# Everything below ran through tests/simt (the kernel sources executed on the host) but not yet on the device: it sits at the end of
# the file so that `pytest -x` reaches it after the suite that was green on the MI355X in GPU calls 13-16.
def test_score_aggregation_modes(dev):
    """MODEL.DYHEAD.SCORE_AGG = MAX / ONEHOT / POWER: mq_align_scores_fwd vs the reference-generated fixture, post-processing vs the oracle"""
    import parity_checks as pc
    _assert(pc.check_score_agg(dev))


# ------------------------------------------------------------------------------------------------ resident-key attention (opt-in)
def _body_resident_attention_kernel(dev, monkeypatch):
    """csrc/attn_resident.hip (MQ_ATTN_RESIDENT=1: S^T formulation; text-sized attentions with all keys resident in LDS, long key
    sequences in chunks of 256) -- written after the round-2 GPU budget was spent and checked through tests/simt only; this is its
    first run on the device."""
    import parity_checks as pc
    import gdino_checks as gc
    monkeypatch.setenv("MQ_ATTN_RESIDENT", "1")
    for cfg in (dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True), dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True, clamp=50000.0, big=True),
                dict(B=1, H=8, D=32, Nq=37, Nk=61), dict(B=2, H=8, D=32, Nq=1, Nk=9), dict(B=3, H=2, D=64, Nq=130, Nk=141, mask=True, kvlen=True),
                dict(B=64, H=12, D=64, Nq=256, Nk=256, mask=True, kvlen=True), dict(B=1, H=4, D=32, Nq=900, Nk=200)):
        _assert(pc.check_attention(dev, **cfg))
    _assert(pc.check_attention_strided(dev))
    _assert(gc.check_attention_qk_mask(dev))
    _assert(pc.check_bert_layer(dev, True))
    # long key sequences: the chunked kernel (GCP pre-select shape with and without key split, decoder self-attention, ragged tails)
    for cfg in (dict(B=1, H=8, D=32, Nq=200, Nk=5577, nsplit=4), dict(B=8, H=8, D=32, Nq=200, Nk=5577), dict(B=2, H=8, D=32, Nq=70, Nk=700, mask=True, nsplit=2),
                dict(B=2, H=2, D=64, Nq=130, Nk=600, mask=True, kvlen=True, clamp=50000.0, big=True), dict(B=1, H=8, D=32, Nq=900, Nk=900),
                dict(B=1, H=2, D=32, Nq=37, Nk=257, nsplit=2), dict(B=1, H=2, D=32, Nq=37, Nk=100, nsplit=3)):
        _assert(pc.check_attention(dev, **cfg))
    _assert(pc.check_pre_select(dev))


def _body_layernorm2_kernel(dev, monkeypatch):
    """csrc/layernorm2.hip (MQ_LN_VARIANT=2: load-batched LayerNorm): the LayerNorm checks and a BERT layer on it, and its outputs next
    to mq_layernorm_fwd's on the same inputs (bit-identical through tests/simt; on the device the two kernels are separate
    compilations of the same expressions, so one rounding step of slack is allowed)."""
    import parity_checks as pc
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(7)
    for rows, C in ((1000, 96), (5, 192), (777, 256), (130, 384), (65, 768), (50, 1536), (7, 3072), (64 * 2048 + 3, 96), (8 * 22400, 256)):
        x32, r32 = torch.randn(rows, C, generator=g) * 2 + 0.5, torch.randn(rows, C, generator=g)
        w, b = (torch.randn(C, generator=g) * 0.1 + 1).half().to(dev), (torch.randn(C, generator=g) * 0.1).half().to(dev)
        for x, res in ((x32.half(), None), (x32, r32.half()), (x32.half(), r32.half()), (x32, r32)):
            x, res = x.to(dev), None if res is None else res.to(dev)
            outs = {}
            for variant in ("1", "2"):
                monkeypatch.setenv("MQ_LN_VARIANT", variant)
                o = ops.layer_norm(x, w, b, 1e-5, residual=res, want_sum=True, want_y32=True)
                outs[variant] = o if isinstance(o, tuple) else (o,)
            for t1, t2 in zip(outs["1"], outs["2"]):
                assert t1.dtype == t2.dtype and t1.shape == t2.shape
                assert torch.allclose(t1.float(), t2.float(), rtol=1e-3 if t1.dtype == torch.float16 else 1e-5, atol=1e-3 if t1.dtype == torch.float16 else 1e-5), (rows, C)
    monkeypatch.setenv("MQ_LN_VARIANT", "2")
    _assert(pc.check_layernorm(dev))
    _assert(pc.check_bert_layer(dev, True))


def _body_offset_conv_v2_kernel(dev, monkeypatch):
    """csrc/conv_small2.hip (MQ_OFFSET_CONV_VARIANT=2: the DyConv offset conv with unconditional in-flight loads): equal to
    mq_conv3x3_nchw32_fwd on the same inputs (same MFMA order: the results must not differ at all), conv and DyConv checks on it."""
    import parity_checks as pc
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(3)
    for B, H, W, C in ((2, 13, 21, 256), (1, 7, 11, 256), (1, 3, 5, 128), (2, 17, 33, 64), (8, 100, 168, 256), (8, 7, 11, 256)):
        x = torch.randn(B, H, W, C, generator=g).half().to(dev)
        w = torch.zeros(32, 9 * C, dtype=torch.float16)
        w[:27] = (torch.randn(27, 9 * C, generator=g) / 48).half()
        w, bias = w.to(dev), torch.randn(27, generator=g).half().to(dev)
        outs = {}
        for variant in ("1", "2"):
            monkeypatch.setenv("MQ_OFFSET_CONV_VARIANT", variant)
            outs[variant] = ops.conv3x3_nchw32(x, w, bias, 27)
        assert torch.equal(outs["1"], outs["2"]), (B, H, W, C)
    monkeypatch.setenv("MQ_OFFSET_CONV_VARIANT", "2")
    _assert(pc.check_conv3x3(dev))
    _assert(pc.check_dyconv(dev))


def _body_offset_conv_group_kernel(dev, monkeypatch):
    """csrc/conv_small3.hip (MQ_OFFSET_CONV_VARIANT=3: the offset conv of every pyramid level in one launch of persistent workgroups that keep
    the weights in registers): against F.conv2d and the per-level kernel, at the benchmark pyramid too; the DyConv block and the tiny
    model on it."""
    import parity_checks as pc
    monkeypatch.setenv("MQ_OFFSET_CONV_VARIANT", "3")
    _assert(pc.check_conv3x3_group(dev))
    _assert(pc.check_dyconv(dev))
    _assert(pc.check_full_model(dev))


def _body_dyconv_epilogue_group(dev, monkeypatch):
    """mq_dyconv_epilogue_group (MQ_DYCONV_EPILOGUE_GROUPED=1: fuse pass + DYReLU coefficients of all levels of a DyConv layer in two launches):
    equal to the per-level launches bit for bit; the DyConv block and the tiny model with the switch on."""
    import parity_checks as pc
    _assert(pc.check_dyconv_epilogue_group(dev))
    monkeypatch.setenv("MQ_DYCONV_EPILOGUE_GROUPED", "1")
    _assert(pc.check_dyconv(dev))
    _assert(pc.check_full_model(dev))


def _body_patch_merge_ln_kernel(dev, monkeypatch):
    """mq_patch_merge_ln_fwd (MQ_PATCH_MERGE_FUSED=1: Swin PatchMerging gather + LayerNorm in one kernel) next to F.pad + cat +
    mq_layernorm_fwd on the same inputs, then Swin + FPN with the switch on."""
    import torch.nn.functional as F
    import parity_checks as pc
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(11)
    for B, H, W, C in ((2, 8, 12, 96), (1, 7, 11, 96), (2, 5, 6, 192), (1, 9, 4, 384), (1, 3, 3, 768), (8, 200, 336, 96), (8, 25, 42, 384)):
        for xd in (torch.float32, torch.float16):
            x = (torch.randn(B, H, W, C, generator=g) * 2 + 0.3).to(xd).to(dev)
            w, b = (torch.randn(4 * C, generator=g) * 0.1 + 1).half().to(dev), (torch.randn(4 * C, generator=g) * 0.1).half().to(dev)
            y = F.pad(x, (0, 0, 0, W % 2, 0, H % 2)) if (H % 2 or W % 2) else x
            y = torch.cat([y[:, 0::2, 0::2], y[:, 1::2, 0::2], y[:, 0::2, 1::2], y[:, 1::2, 1::2]], -1)
            ref = ops.layer_norm(y.reshape(B, -1, 4 * C).contiguous(), w, b, 1e-5)
            got = ops.patch_merge_ln(x, w, b, 1e-5)
            assert got.shape == ref.shape and torch.allclose(got.float(), ref.float(), rtol=1e-3, atol=1e-3), (B, H, W, C, xd)
    monkeypatch.setenv("MQ_PATCH_MERGE_FUSED", "1")
    pc._CACHE.clear()
    _assert(pc.check_swin_fpn(dev))
    pc._CACHE.clear()


def _body_fpn_convs_through_the_grouped_dcn_kernel(dev, monkeypatch):
    """MQ_FPN_VIA_DCN=1: the FPN output convs as one grouped launch of the fused DCNv2 kernel with zero offsets (== plain 3x3 conv)"""
    import parity_checks as pc
    from mq_det_amd.modeling import pipeline
    spec, sd, cfg, model, P = pc.tiny(dev)
    g = torch.Generator().manual_seed(5)
    feats = [torch.randn(2, h, w, c, generator=g).half().to(dev) for (h, w), c in (((100, 168), 192), ((50, 84), 384), ((25, 42), 768))]
    monkeypatch.setenv("MQ_FPN_VIA_DCN", "0")
    ref = pipeline.fpn_forward(P, feats)
    monkeypatch.setenv("MQ_FPN_VIA_DCN", "1")
    got = pipeline.fpn_forward(P, feats)
    for a, b in zip(got, ref):
        assert a.shape == b.shape and float((a.float() - b.float()).abs().max()) <= 2e-3 * max(1.0, float(b.float().abs().max()))
    pc._CACHE.clear()
    _assert(pc.check_swin_fpn(dev))
    pc._CACHE.clear()


def _body_nms_early_stop(dev, monkeypatch):
    """mq_ml_nms_topk (MQ_NMS_EARLY_STOP=1): the max_keep best survivors equal mq_ml_nms's, nothing else is kept; post-processing on it"""
    import parity_checks as pc
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(13)
    B, N = 8, 5000
    xy = torch.rand(B, N, 2, generator=g) * 600
    boxes = torch.cat([xy, xy + 15 + torch.rand(B, N, 2, generator=g) * 60], -1).contiguous().to(dev)
    labels = torch.randint(1, 41, (B, N), generator=g, dtype=torch.int32).to(dev)
    nvalid = torch.tensor([5000, 4100, 3000, 700, 130, 64, 1, 0], dtype=torch.int32).to(dev)
    monkeypatch.setenv("MQ_NMS_EARLY_STOP", "0")
    full = ops.ml_nms(boxes, labels, nvalid, 0.6).cpu()
    monkeypatch.setenv("MQ_NMS_EARLY_STOP", "1")
    for K in (1, 100, 300, 6000):
        part = ops.ml_nms(boxes, labels, nvalid, 0.6, max_keep=K).cpu()
        for b in range(B):
            kf, kp = full[b].nonzero().flatten(), part[b].nonzero().flatten()
            n = min(K, len(kf))
            assert len(kp) >= n and torch.equal(kp[:n], kf[:n]) and bool((part[b] <= full[b]).all()), (K, b)
    _assert(pc.check_post_golden(dev))


class _Env:
    """monkeypatch.setenv for the isolated bodies (plain os.environ: the process ends with the body); the product reads its kernel
    selection once (ops.configure), so it is re-read after every change"""
    @staticmethod
    def setenv(k, v):
        os.environ[k] = v
        from mq_det_amd import ops
        ops.configure()


def _isolated(body, timeout=900):
    """The bodies above drive kernels that have NEVER run on a device (written after the last GPU call of round 2; checked through
    tests/simt: parity, guard pages, permuted schedules).  A wrong address on the device is not a Python exception but a memory fault
    that kills the process -- so each body runs in a process of its own: a fault there fails one test, not the session."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), body], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{body}: rc {r.returncode}\n{(r.stdout + r.stderr)[-3000:]}"


@pytest.mark.parametrize("body", ["resident_attention_kernel", "layernorm2_kernel", "offset_conv_v2_kernel", "offset_conv_group_kernel", "dyconv_epilogue_group", "patch_merge_ln_kernel",
                                  "fpn_convs_through_the_grouped_dcn_kernel", "nms_early_stop"])
def test_opt_in_kernel(dev, body):
    _isolated(body)


def _body_alternate_kernel_selection(dev, monkeypatch):
    """Every operator that has two implementations, on the one that is NOT the default (ops.KERNEL_DEFAULTS): the streaming attention
    kernel, LayerNorm v1, the guarded-load offset conv, pad + cat patch merging, per-conv FPN launches, the full NMS sweep, the first
    Swin MLP kernel, bmm + mq_align_scores_fwd instead of mq_align_fused_fwd, DYReLU as a pass of its own, the first VLFuse image-side
    kernel for every caption length -- the tiny full model end to end against the oracle."""
    import parity_checks as pc
    from mq_det_amd import ops
    for k in ("ATTN_RESIDENT", "PATCH_MERGE_FUSED", "FPN_VIA_DCN", "NMS_EARLY_STOP", "ALIGN_FUSED", "DYRELU_IN_LN", "SWIN_QKV_FUSED", "FPN_TOPDOWN_FUSED"):
        monkeypatch.setenv("MQ_" + k, "0")
    for k in ("LN_VARIANT", "OFFSET_CONV_VARIANT", "SWIN_MLP_VARIANT", "VLFUSE_I2T_VARIANT"):
        monkeypatch.setenv("MQ_" + k, "1")
    assert ops.KERNELS["ALIGN_FUSED"] == 0 and ops.KERNELS["SWIN_MLP_VARIANT"] == 1
    _assert(pc.check_full_model(dev))
    _assert(pc.check_swin_fpn(dev))


def test_alternate_kernel_selection(dev):
    _isolated("alternate_kernel_selection")


# ------------------------------------------------------------------------------------------------ bf16 operands (configs[3])
@pytest.fixture()
def bf16():
    """The *_bf16 entry points: the same kernel sources compiled with bf16 operands (csrc/common.h, -DMQ_BF16).  Every check runs
    with inputs / weights rounded to bf16 and 8x the fp16 tolerance (3 fewer significant bits), tests/parity_checks.py."""
    import parity_checks as pc
    pc.use_dtype(torch.bfloat16)
    yield pc
    pc.use_dtype(torch.float16)


# the complete bf16 list runs by default since round 3 (all 17 passed on the MI355X in GPU call 1: + ~65 s)
@pytest.mark.parametrize("name", ["check_attention_strided", "check_window_attention", "check_vlfuse_kernels", "check_dcn", "check_layernorm",
                                  "check_swin_mlp", "check_full_model", "check_gcp_block", "check_pre_select", "check_vl_fuse", "check_dyconv",
                                  "check_conv3x3", "check_post_golden", "check_roi_align", "check_msdeform_attn", "check_align_fused",
                                  "check_attention_text", "check_bert_attn_qkv", "check_gcp_attn_fused", "check_patch_embed", "check_bert_clamp_fused"])
def test_bf16_block(dev, bf16, name):
    _assert(getattr(bf16, name)(dev))


def test_bf16_mq_glip_l_family(dev, bf16):
    """BASELINE configs[3] as named: MQ-GLIP-L (Swin-L window 12, tiny depth) on bf16 MFMA."""
    _assert(bf16.check_window_attention(dev, large=True))
    _assert(bf16.check_full_model(dev, large=True))


def test_bf16_mq_glip_l_benchmark_configuration_parity(dev, bf16):
    """BASELINE configs[3] as named -- "MQ-GLIP-L ... bf16 MFMA" -- at FULL depth on 800x1333, gated by the bf16-operand floor."""
    _assert(bf16.check_benchmark_config(dev, "long", ((800, 1333),), family="l"))


def test_bf16_groundingdino(dev, bf16):
    import gdino_checks as gc
    _assert(gc.check_msdeform_attn_q(dev))
    _assert(gc.check_gdino_model(dev, vq=True))



# ------------------------------------------------------------------------------------------------ fp32 operands: the precise mode on the device
@pytest.fixture()
def f32():
    """MODEL.COMPUTE_DTYPE = "float32", the SPLIT-PRECISE mode: the *_f32 entry points (the same kernel sources, every operand a float carried as
    hi + lo / 2^11 through three fp16 MFMAs, csrc/common.h) + fp32 library GEMMs, ON THE MI355X.  Every row is gated at the north-star tolerance: max|err| <= 1e-3 of the
    reference's range AND no element outside atol = rtol = 1e-3 (parity_checks._stat)."""
    import parity_checks as pc
    from mq_det_amd import ops
    prev = os.environ.get("MQ_F32_OPERANDS")
    os.environ["MQ_F32_OPERANDS"] = "1"
    ops.configure()
    pc.use_dtype(torch.float32)
    yield pc
    pc.use_dtype(torch.float16)
    if prev is None:
        os.environ.pop("MQ_F32_OPERANDS", None)
    else:
        os.environ["MQ_F32_OPERANDS"] = prev
    ops.configure()


def _assert_f32(res):
    res = res if isinstance(res, list) else [res]
    _assert(res)
    # set-valued rows (matched detections, the two-stage top-k overlap, replay-vs-eager matches) carry a fraction, not a normalised error
    loose = [r for r in res if r["tol"] > 1e-3 and not any(t in r["name"] for t in ("detections", "two-stage top-", "HIP-graph replay", "replay vs eager forward", "vs the B = 1 forward"))]
    assert not loose, "rows gated above 1e-3 in the precise mode: " + ", ".join(r["name"] for r in loose)
    viol = [r for r in res if not r.get("elem_ok", True)]
    assert not viol, "elements outside atol = rtol = 1e-3: " + ", ".join(f"{r['name']} ({r['elem_viol_frac']:.1e})" for r in viol)


@pytest.mark.parametrize("name", ["check_attention_strided", "check_window_attention", "check_swin_fpn", "check_vlfuse_kernels", "check_dcn", "check_layernorm",
                                  "check_swin_mlp", "check_gcp_block", "check_pre_select", "check_vl_fuse", "check_dyconv", "check_conv3x3",
                                  "check_align_fused", "check_attention_text", "check_bert_attn_qkv", "check_gcp_attn_fused", "check_patch_embed", "check_bert_clamp_fused",
                                  "check_roi_align", "check_extract_query"])
def test_f32_block(dev, f32, name):
    _assert_f32(getattr(f32, name)(dev))


@pytest.mark.parametrize("clamp", [False, True])
def test_f32_bert_layer(dev, f32, clamp):
    _assert_f32(f32.check_bert_layer(dev, clamp))


def test_f32_full_model(dev, f32):
    """The tiny-depth full model of smoke(): Swin -> FPN -> BERT + GCP -> fusion layers -> heads -> class scores, every stage at 1e-3."""
    _assert_f32(f32.check_full_model(dev))


def test_f32_fusion_layer_at_the_benchmark_geometry(dev, f32):
    """One fusion layer (VLFuse both ways, clamped BERT layer, DyConv / DCNv2) on the 22 400 pyramid tokens of an 800 x 1333 image, 141 live
    text tokens: 1e-3 at every output, on the device."""
    _assert_f32(f32.check_fusion_layer(dev))


def test_f32_benchmark_configuration_b8_graph_replay(dev, f32):
    """"make 1e-3 a property of something you benchmark at B = 8".  THE configuration `bench.py --dtype f32` times -- B = 8 images
    800 x 1333, 141-token caption, split-precise kernels, the forward replayed from a HIP graph: every stage row of batch item 0 within 1e-3 of the
    oracle (no element outside atol = rtol = 1e-3), the replayed detections of item 0 against the oracle's, replay vs eager for every image, items 3
    and 7 against B = 1 forwards."""
    _assert_f32(f32.check_benchmark_b8_graph(dev))


def test_f32_mq_glip_l_family(dev, f32):
    """The split-precise mode on the MQ-GLIP-L family (Swin-L: window 12 = 144-token windows padded to 160, widths 192 ... 1536, 8 fusion layers in the
    full model; here the tiny-depth model of test_mq_glip_l_family): window attention, Swin + FPN and the whole forward at 1e-3."""
    _assert_f32(f32.check_window_attention(dev, large=True))
    _assert_f32(f32.check_swin_fpn(dev, large=True))
    _assert_f32(f32.check_full_model(dev, large=True))


def test_f32_groundingdino(dev, f32):
    """the split-precise mode for MQ-GroundingDINO (groundingdino.py:438-661; the reference forces fp32 inside MSDeformAttn,
    ms_deform_attn.py:330-336): sampling kernels, the shallow model and the FULL-DEPTH model at 800 x 1333 (6 + 6 layers, 900 queries, 40 classes x
    5 vision queries) -- every stage within 1e-3 of the oracle's range, no element outside atol = rtol = 1e-3."""
    import gdino_checks as gc
    _assert_f32(gc.check_msdeform_attn_q(dev))
    _assert_f32(gc.check_attention_qk_mask(dev))
    _assert_f32(gc.check_vlfuse_heads_mask(dev))
    _assert_f32(gc.check_gdino_model(dev, vq=True, graph=False))
    _assert_f32(gc.check_gdino_benchmark_config(dev))


def test_f32_benchmark_configuration_parity(dev, f32):
    """THE north-star statement: full-depth MQ-GLIP-T on an 800 x 1333 image with the 141-token caption against the fp32 oracle -- every
    stage of the ladder (Swin, FPN, language backbone, each of the 6 fusion layers, box / centerness / alignment logits, class scores)
    within 1e-3, detections matched -- on the MI355X, with the kernels of the product compiled for fp32 operands."""
    _assert_f32(f32.check_benchmark_config(dev, "long", ((800, 1333),)))


if __name__ == "__main__":                       # python tests/test_gpu_parity.py <body>: one isolated body (see _isolated)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _dev = torch.device("cuda:0")
    from mq_det_amd import ops as _ops
    _ops.load_library()
    globals()["_body_" + sys.argv[1]](_dev, _Env)
    torch.cuda.synchronize()
    print("body ok:", sys.argv[1])
