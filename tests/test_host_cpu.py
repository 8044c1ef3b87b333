"""CPU tests: C-ABI library loads and exports every symbol include/mqdet_hip.h declares (no compute calls),
boundary containers, config, tokenizer glue, state_dict naming contract, fail-loud behaviour without a GPU."""
import os
import re
import subprocess
import sys
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_match_header():
    from mq_det_amd import build, ops
    build.build()
    hdr = open(os.path.join(ROOT, "include", "mqdet_hip.h")).read()
    twins = set(re.findall(r"^MQ_BF16_TWIN\((mq_[a-z0-9_]+)\)", hdr, re.M))
    assert twins == set(ops.BF16_TWINS), twins ^ set(ops.BF16_TWINS)
    twins32 = set(re.findall(r"^MQ_F32_TWIN\((mq_[a-z0-9_]+)\)", hdr, re.M))
    assert twins32 == set(ops.F32_TWINS), twins32 ^ set(ops.F32_TWINS)
    declared = set(re.findall(r"\b(mq_[a-z0-9_]+)\s*\(", hdr)) | {n + "_bf16" for n in twins} | {n + "_f32" for n in twins32}
    assert declared == set(ops.EXPORTS), declared ^ set(ops.EXPORTS)
    lib = ops.load_library()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mq_abi_version() == ops.EXPECTED_ABI == 31
    assert lib.mq_attn_workspace_bytes(2, 8, 256, 256, 4) == 4 * 2 * 8 * 256 * 258 * 4
    assert lib.mq_ml_nms_workspace_bytes(2, 130) == 2 * 130 * 3 * 8


def test_ops_fail_loudly_without_gpu():
    from mq_det_amd import ops
    q = torch.zeros(1, 8, 64, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.attention(q, q, q.transpose(1, 2).contiguous(), 1, 64)


def test_model_refuses_cpu_and_training():
    from mq_det_amd import get_cfg, build_detection_model
    cfg = get_cfg()
    cfg.MODEL.SWINT.DEPTHS = (2, 2, 2, 2)
    cfg.MODEL.LANGUAGE_BACKBONE.NUM_HIDDEN_LAYERS = 2
    cfg.MODEL.LANGUAGE_BACKBONE.QV_START = 1
    cfg.MODEL.LANGUAGE_BACKBONE.BERT_VOCAB_SIZE = 1100
    cfg.MODEL.DYHEAD.NUM_CONVS = 1
    model = build_detection_model(cfg, tokenizer=object())
    with pytest.raises(RuntimeError, match="MI355X only"):
        model(torch.zeros(1, 3, 64, 64), captions=["a"], positive_map={1: [1]})
    with pytest.raises(NotImplementedError):
        model.train()
    assert model.backbone.body is not None and model.backbone.fpn is not None and model.rpn.head is not None


def test_unsupported_config_is_rejected_by_name():
    from mq_det_amd import get_cfg, build_detection_model
    cfg = get_cfg()
    cfg.MODEL.SWINT.DEPTHS = (2, 2, 2, 2)
    cfg.MODEL.LANGUAGE_BACKBONE.NUM_HIDDEN_LAYERS = 2
    cfg.MODEL.LANGUAGE_BACKBONE.QV_START = 1
    cfg.MODEL.LANGUAGE_BACKBONE.BERT_VOCAB_SIZE = 1100
    cfg.MODEL.DYHEAD.NUM_CONVS = 1
    cfg.MODEL.DYHEAD.SCORE_AGG = "POWER"           # the reference: POWER exists in the MDETR-style aggregation only
    cfg.TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM = -1
    model = build_detection_model(cfg, tokenizer=object())
    with pytest.raises(NotImplementedError, match="SCORE_AGG"):
        model._validate_config()
    cfg.TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM = 3000
    model._validate_config()
    cfg.MODEL.DYHEAD.SCORE_AGG = "MEDIAN"
    with pytest.raises(NotImplementedError, match="SCORE_AGG"):
        model._validate_config()
    for agg in ("MAX", "ONEHOT", "MEAN"):
        cfg.MODEL.DYHEAD.SCORE_AGG = agg
        model._validate_config()
    cfg.MODEL.LANGUAGE_BACKBONE.PAD_MAX = False            # accepted: same detections (tokenize pads to MAX_QUERY_LEN, masked)
    model._validate_config()
    cfg.MODEL.LANGUAGE_BACKBONE.MAX_QUERY_LEN = 300
    with pytest.raises(NotImplementedError, match="MAX_QUERY_LEN"):
        model._validate_config()


def test_state_dict_names_match_reference_contract():
    """Same keys as the oracle generator, whose keys load strict=True into the reference's own classes
    (oracle/gen_golden.py)."""
    from oracle import glip_t_spec
    from oracle.weights import make_state_dict
    from mq_det_amd import get_cfg
    from mq_det_amd.modeling.params import param_specs
    cfg = get_cfg()
    cfg.MODEL.DYHEAD.NUM_CLASSES = 1204
    ours = {n: tuple(s) for n, s, _ in param_specs(cfg)}
    ref = {k: tuple(v.shape) for k, v in make_state_dict(glip_t_spec(), 0).items()}
    assert ours == ref


def test_boxlist_and_image_list():
    from mq_det_amd import BoxList, to_image_list, cat_boxlist
    b = BoxList(torch.tensor([[0., 0., 9., 9.], [5., 5., 200., 300.]]), (100, 50))
    b.add_field("scores", torch.tensor([0.5, 0.7]))
    assert b.area().tolist() == [100.0, 196.0 * 296.0]
    c = b.clip_to_image(remove_empty=False)
    assert c.bbox[1].tolist() == [5.0, 5.0, 99.0, 49.0]
    assert b.convert("xywh").bbox[0].tolist() == [0.0, 0.0, 10.0, 10.0]
    assert len(cat_boxlist([b, b])) == 4 and len(b[torch.tensor([True, False])]) == 1
    il = to_image_list([torch.ones(3, 30, 40), torch.ones(3, 33, 20)], 32)
    assert il.tensors.shape == (2, 3, 64, 64) and il.image_sizes == [(30, 40), (33, 20)]
    assert float(il.tensors[1, :, 33:].sum()) == 0


def test_cfg_merge_and_freeze(tmp_path):
    from mq_det_amd import get_cfg
    cfg = get_cfg()
    y = tmp_path / "c.yaml"
    y.write_text("MODEL:\n  RPN:\n    ASPECT_RATIOS: (1.0,)\n  DYHEAD:\n    NUM_CONVS: 8\nTEST:\n  CHUNKED_EVALUATION: 40\n")
    cfg.merge_from_file(str(y))
    cfg.merge_from_list(["MODEL.ATSS.DETECTIONS_PER_IMG", "300"])
    assert cfg.MODEL.DYHEAD.NUM_CONVS == 8 and cfg.MODEL.RPN.ASPECT_RATIOS == (1.0,) and cfg.MODEL.ATSS.DETECTIONS_PER_IMG == 300
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.MODEL.DEVICE = "cpu"
    assert cfg.clone().MODEL.DYHEAD.NUM_CONVS == 8


def test_tokenizer_and_positive_map():
    from transformers import AutoTokenizer
    from mq_det_amd.utils.tokenizer import build_synthetic_tokenizer, synthetic_caption, positive_map_from_spans
    tk = AutoTokenizer.from_pretrained(build_synthetic_tokenizer(tempfile.mkdtemp(), size=3000))
    cap, spans = synthetic_caption(40, words=(1,))                     # round-1 caption: one token per class
    pm = positive_map_from_spans(tk, cap, spans, list(range(1, 41)))
    assert len(pm) == 40 and pm[1] == [1] and pm[2] == [3]
    t = tk([cap], max_length=256, padding="max_length", return_tensors="pt", truncation=True)
    assert t["input_ids"].shape == (1, 256) and t["input_ids"][0, 0] == 101 and int(t["attention_mask"].sum()) == 81
    cap, spans = synthetic_caption(40)                                  # default: 1-4 word names, LVIS-chunk length
    pm = positive_map_from_spans(tk, cap, spans, list(range(1, 41)))
    assert len(pm) == 40 and pm[1] == [1] and pm[2] == [3, 4] and pm[4] == [10, 11, 12, 13]
    t = tk([cap], max_length=256, padding="max_length", return_tensors="pt", truncation=True)
    assert 120 <= int(t["attention_mask"].sum()) <= 200


def test_live_row_compaction_policy_and_raw_padding():
    """detector._live_len: 16 ceil(live / 16) text positions (never more than the tokenizer's T; all T without a bound or with
    MODEL.LANGUAGE_BACKBONE.COMPACT_TEXT off); _pad_raw_text brings the per-token tensors of a raw forward back to [.., T, ..] without touching the
    operands postprocess() may be called on again."""
    from mq_det_amd import get_cfg
    from mq_det_amd.modeling.detector import GeneralizedVLRCNN_New
    m = GeneralizedVLRCNN_New.__new__(GeneralizedVLRCNN_New)
    m.cfg = get_cfg()
    assert [m._live_len(256, kv) for kv in (0, 1, 16, 17, 141, 250, 256, 300)] == [256, 16, 16, 32, 144, 256, 256, 256]
    m.cfg.MODEL.LANGUAGE_BACKBONE.COMPACT_TEXT = False
    assert m._live_len(256, 141) == 256
    raw = {"lang": {"hidden": torch.ones(2, 144, 8), "hidden32": None, "embedded": torch.ones(2, 144, 8), "masks": torch.ones(2, 144), "key_bias": torch.zeros(2, 144)},
           "head": {"hidden": torch.ones(2, 144, 8), "tbias": torch.ones(2, 144), "dot": [torch.ones(2, 5, 144)]},
           "head_trace": [{"bert_hidden": torch.ones(2, 144, 8)}]}
    out = GeneralizedVLRCNN_New._pad_raw_text(raw, 256)
    assert out["lang"]["hidden"].shape == (2, 256, 8) and float(out["lang"]["hidden"][:, 144:].abs().sum()) == 0 and out["lang"]["masks"].shape == (2, 256)
    assert out["head"]["hidden"].shape == (2, 256, 8) and out["head_trace"][0]["bert_hidden"].shape == (2, 256, 8)
    assert out["head"]["tbias"].shape == (2, 144) and out["head"]["dot"][0].shape == (2, 5, 144)


def test_replay_input_copies_skip_only_unmodified_small_tensors():
    """GraphRunner._tree_copy_: a small input that the detector itself memoised (graph_runner.memoised: token ids, masks, selections, ...), that is
    the very tensor object of the previous replay and unmodified (version counter) is not copied again; a modified one, another object, a large one
    -- and ANY tensor that is not a registered memo (a caller's tensor may have been written through data_ptr / .data without a version
    bump; inference tensors have no counter) -- is."""
    from mq_det_amd.modeling.graph_runner import GraphRunner, memoised, is_memoised
    small, big = memoised(torch.arange(8.0)), torch.zeros(GraphRunner.SKIP_COPY_MAX_NUMEL + 1)
    dst = {"a": torch.zeros(8), "b": [torch.zeros_like(big)]}
    seen = {}
    GraphRunner._tree_copy_(dst, {"a": small, "b": [big]}, seen)
    assert torch.equal(dst["a"], small) and ("a",) in seen and ("b", 0) not in seen
    dst["a"].zero_()                                                   # (stands for: the static buffer already holds the value -- a skipped copy leaves it alone)
    GraphRunner._tree_copy_(dst, {"a": small, "b": [big]}, seen)
    assert float(dst["a"].abs().sum()) == 0                            # same memoised object, same version: skipped
    small.add_(1)                                                      # in-place write bumps the version counter
    GraphRunner._tree_copy_(dst, {"a": small, "b": [big]}, seen)
    assert torch.equal(dst["a"], small)
    other = small.clone()
    dst["a"].zero_()
    GraphRunner._tree_copy_(dst, {"a": other, "b": [big]}, seen)
    assert torch.equal(dst["a"], other)                                # another object with the same content: copied
    big[0] = 5.0
    GraphRunner._tree_copy_(dst, {"a": other, "b": [big]}, seen)
    assert float(dst["b"][0][0]) == 5.0                                # large tensors: always
    # a caller's own small tensor: copied on EVERY replay, also when nothing visible changed (a .data write does not bump the counter)
    mine = torch.arange(8.0)
    assert not is_memoised(mine)
    GraphRunner._tree_copy_(dst, {"a": mine, "b": [big]}, seen)
    mine.data[0] = 77.0
    GraphRunner._tree_copy_(dst, {"a": mine, "b": [big]}, seen)
    assert float(dst["a"][0]) == 77.0
    with torch.inference_mode():                                       # inference tensors have no version counter: must not raise, must copy
        inf = torch.arange(8.0) + 3
        GraphRunner._tree_copy_(dst, {"a": inf, "b": [big]}, seen)
        GraphRunner._tree_copy_(dst, {"a": memoised(inf), "b": [big]}, seen)
    assert float(dst["a"][0]) == 3.0
    # nested memos (the query-bank selection is a tuple)
    pair = memoised((torch.zeros(2), {"k": torch.ones(2)}))
    assert is_memoised(pair[0]) and is_memoised(pair[1]["k"])


def test_b_fragment_order_is_the_headers_formula():
    """ops.pack_b_fragments = the layout include/mqdet_hip.h states (MFMA B-fragment order), and unpack_b_fragments inverts it."""
    import torch
    from mq_det_amd import ops
    N, K = 48, 96
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K)
    p = ops.pack_b_fragments(w)
    assert p.shape == (N // 16, K // 32, 64, 8) and p.is_contiguous()
    flat = p.reshape(-1)
    for n, k in ((0, 0), (5, 9), (17, 33), (47, 95), (16, 31), (31, 64)):
        i = ((n // 16) * (K // 32) + k // 32) * 512 + ((k % 32) // 8 * 16 + n % 16) * 8 + k % 8
        assert flat[i] == w[n, k]
    # lane l of a wave reads 8 consecutive elements at 8 l: row n % 16 = l % 16, k offset 8 (l / 16) -- the operand layout of v_mfma_f32_16x16x32
    assert torch.equal(p[1, 2, 37], w[16 + 37 % 16, 64 + 8 * (37 // 16):64 + 8 * (37 // 16) + 8])
    assert torch.equal(ops.unpack_b_fragments(p), w)


def test_fused_text_kernel_size_policy():
    """KERNELS[...] = 1: the fused text kernels over the measured range (mq_bert_attn_qkv_fwd up to FUSED_BERT_MAX_WORKGROUPS (batch item, head)
    pairs, mq_gcp_attn_fwd up to FUSED_TEXT_MAX_ROWS text rows: B = 64 of the benchmark caption); = 2 always."""
    from mq_det_amd import ops
    saved = dict(ops.KERNELS)
    try:
        ops.KERNELS["BERT_ATTN_QKV_FUSED"], ops.KERNELS["GCP_ATTN_FUSED"] = 1, 1
        assert ops.bert_attention_qkv_fits(144, 768, 12, None, batch=8) and ops.bert_attention_qkv_fits(144, 768, 12, None, batch=64)
        assert not ops.bert_attention_qkv_fits(144, 768, 12, None, batch=65) and not ops.bert_attention_qkv_fits(144, 1024, 16, None, batch=8)
        x8, x64 = torch.zeros(8, 144, 768), torch.zeros(65, 144, 768)
        idx = torch.zeros(8, 144, 5, dtype=torch.int32)
        assert ops.gcp_attention_fits(x8, idx, policy=True) and not ops.gcp_attention_fits(x64, idx, policy=True) and ops.gcp_attention_fits(x64, idx)
        assert ops.gcp_attention_fits(torch.zeros(64, 144, 768), idx, policy=True)
        assert not ops.gcp_attention_fits(x8, torch.zeros(8, 144, 9, dtype=torch.int32))                 # more than 8 slots: the unfused path
        # the split-precise mode: the policy takes the unfused launches (the fused kernels still split inside mfma16: GPU call 4 of round 6) ...
        ops.KERNELS["F32_OPERANDS"] = 1
        assert not ops.bert_attention_qkv_fits(144, 768, 12, None, batch=8) and not ops.gcp_attention_fits(x8, idx, policy=True)
        assert ops.bert_attention_qkv_fits(144, 768, 12, None) and ops.gcp_attention_fits(x8, idx)          # ... the kernels themselves still take the shapes
        ops.KERNELS["F32_OPERANDS"] = 0
        ops.KERNELS["BERT_ATTN_QKV_FUSED"], ops.KERNELS["GCP_ATTN_FUSED"] = 2, 2
        assert ops.bert_attention_qkv_fits(144, 768, 12, None, batch=64) and ops.gcp_attention_fits(x64, idx, policy=True)
    finally:
        ops.KERNELS.clear()
        ops.KERNELS.update(saved)


def test_kernel_selection_is_per_host_thread():
    """ops.KERNELS is the LIVE selection that activate() rewrites at the top of every forward.  It is one table per host thread:
    a model driven from another thread (other cfg.MODEL.KERNELS, e.g. the split-precise mode) cannot flip this thread's selection between two
    launches; a fresh thread starts from the import-time table (defaults <- environment)."""
    import threading
    from mq_det_amd import ops
    base = dict(ops.KERNELS)
    assert base and ops.KERNELS.copy() == base and len(ops.KERNELS) == len(base) and set(ops.KERNELS) == set(base)
    seen, go, done = {}, threading.Event(), threading.Event()

    def other():
        seen["start"] = dict(ops.KERNELS)
        ops.activate(dict(base, F32_OPERANDS=1, LN_VARIANT=1))
        seen["mine"] = (ops.f32_operands(), ops.KERNELS["LN_VARIANT"])
        go.set()
        done.wait(10)
        seen["still"] = (ops.f32_operands(), ops.KERNELS["LN_VARIANT"])
    t = threading.Thread(target=other)
    t.start()
    assert go.wait(10)
    assert ops.f32_operands() == base["F32_OPERANDS"] and ops.KERNELS["LN_VARIANT"] == base["LN_VARIANT"]        # untouched here
    ops.activate(dict(base, LN_VARIANT=2))
    done.set()
    t.join()
    assert seen["start"] == base and seen["mine"] == (1, 1) and seen["still"] == (1, 1)
    ops.activate(base)
    assert dict(ops.KERNELS) == base


def test_gloo_world2_detection_gather():
    """N > 1 path on CPU: 2 processes, gloo, fixed-shape all-gather of detections + shard ranges."""
    script = os.path.join(ROOT, "tests", "_gloo_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29531", script],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("GATHER_OK") == 2


def test_rccl_worker_dry_run_over_gloo():
    """tests/_rccl_worker.py is what the GPU suite starts over RCCL (one rank on a one-GPU box, two on the first box that has two): the same script,
    CPU tensors, gloo, two ranks -- the detection gather (also one step behind) and the evaluator exchange against per-rank expectations."""
    script = os.path.join(ROOT, "tests", "_rccl_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", MQ_WORKER_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29543", script],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("RCCL_GATHER_OK") == 2


def test_bench_gpus_flag_starts_the_ranks():
    """`python bench.py --gpus 2` (no launcher around it) starts 2 ranks itself; the line says n_gpus = 2 and the one
    fixed-shape gather of the data path ran across them (--dry-launch: no kernels, gloo).  A rank count that differs from --gpus is
    refused with a non-zero status instead of reporting a number for the wrong world size."""
    import json
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-launch", "--steps", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    d = lines[0]
    assert d["n_gpus"] == 2 and d["comm"]["ranks"] == 2 and d["gather_ok"] and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    # the driver's explicit form, same result
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29537", bench, "--gpus", "2", "--dry-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert [json.loads(ln)["n_gpus"] for ln in r.stdout.splitlines() if ln.startswith("{")] == [2]
    # the world size of BASELINE.json configs[2] (8 ranks, gloo, no kernels), with the gather one step behind the "forward" (--overlap-gather)
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--dry-launch", "--overlap-gather", "--steps", "3"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(d) == 1 and d[0]["n_gpus"] == 8 and d[0]["gather_ok"] and d[0]["overlap_gather"] and d[0]["config"]["global_batch"] == 64
    # mismatch: 1 rank in the environment, 2 asked for
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-launch"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in r.stderr and "{" not in r.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="compares with reference code executed in place")
def test_gloo_world2_evaluator_gather_matches_reference():
    """per-category top-k accumulation + cross-rank gather as tensors == the reference's LvisEvaluatorFixedAP
    (update / _merge_lists / synchronize_between_processes over its pickling all_gather), 2 gloo processes."""
    script = os.path.join(ROOT, "tests", "_gloo_eval_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", script],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("EVAL_GATHER_OK") == 2


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="compares with reference code executed in place")
def test_evaluator_update_boxlist_matches_reference_prepare():
    """a detection handed over as a BoxList must reach the LVIS API with the bbox `LvisEvaluatorFixedAP.prepare` builds
    (lvis_eval.py:810-835 via convert_to_xywh :998-1000: no legacy +1), from the (image id, dict) pairs of engine/inference.py:643-648."""
    import torch
    from mq_det_amd.evaluation import FixedAPAccumulator
    from mq_det_amd.structures import BoxList
    from oracle import _refload
    ev = _refload.reference_classes("maskrcnn_benchmark/data/datasets/evaluation/lvis/lvis_eval.py", ["LvisEvaluatorFixedAP"],
                                    ["_merge_lists", "convert_to_xywh"], {"LVIS": object})
    ref = ev["LvisEvaluatorFixedAP"](gt=None, topk=50)
    acc = FixedAPAccumulator(topk=50)
    g = torch.Generator().manual_seed(5)
    preds = []
    for img in (11, 12, 13):
        n = 9
        xy = torch.rand(n, 2, generator=g) * 300
        wh = torch.rand(n, 2, generator=g) * 200 + 0.25
        boxes = torch.cat([xy, xy + wh], 1)
        labels, scores = torch.randint(1, 4, (n,), generator=g), torch.rand(n, generator=g)
        preds.append((img, {"scores": scores, "labels": labels, "boxes": boxes}))
        bl = BoxList(boxes, (640, 480), mode="xyxy")
        bl.add_field("scores", scores)
        bl.add_field("labels", labels)
        acc.update_boxlist(img, bl)
    preds.append((14, {"scores": torch.zeros(0), "labels": torch.zeros(0, dtype=torch.long), "boxes": torch.zeros(0, 4)}))
    empty = BoxList(torch.zeros(0, 4), (640, 480), mode="xyxy")
    empty.add_field("scores", torch.zeros(0))
    empty.add_field("labels", torch.zeros(0, dtype=torch.long))
    acc.update_boxlist(14, empty)
    ref.update(preds)                                     # update() -> prepare() -> convert_to_xywh
    mine = acc.by_cat()
    assert set(mine) == set(ref.by_cat)
    for cat, anns in ref.by_cat.items():
        a = sorted((x["image_id"], round(x["score"], 6), tuple(round(v, 4) for v in x["bbox"])) for x in anns)
        b = sorted((x["image_id"], round(x["score"], 6), tuple(round(v, 4) for v in x["bbox"])) for x in mine[cat])
        assert a == b, (cat, a[:2], b[:2])


def test_bench_roofline_records_are_per_kernel_and_read_the_newest_pmc_file():
    """`roofline` = the SINGLE hand-written kernel with the most time per step (the two-kernel VLFuse record is kept for
    continuity but flagged and never chosen); PMC traffic comes from the newest profiles/r0N_pmc_traffic.json."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    pmc = bench._pmc()
    assert pmc.get("_file", "").startswith("profiles/r06") and pmc.get("dcn_igemm8_kernel", 0) > 1e8          # the newest round's file
    kern = {"dcnv2_fused": (18, 9.9, 0), "vlfuse_i2t_n22400_t256": (18, 6.0, 0), "vlfuse_t2i_n22400_t256_s7": (18, 4.5, 0),
            "swin_mlp_c384": (18, 3.5, 10 ** 9)}
    roofs = bench.kernel_rooflines(kern, 3, 8, 141)
    pair = [r for r in roofs if r.get("pair")]
    assert len(pair) == 1 and abs(pair[0]["ms_per_step"] - 3.5) < 1e-6
    single = [r for r in roofs if r["bound"] == "mfma" and not r.get("pair")]
    best = max(single, key=lambda r: r["ms_per_step"])
    assert best["kernel"].startswith("dcn_igemm8_kernel") and best["traffic"] == pmc["dcn_igemm8_kernel"]
    assert {r["kernel"].split(" ")[0] for r in single} >= {"dcn_igemm8_kernel", "vlfuse_i2t_kernel", "vlfuse_t2i_kernel", "swin_mlp2_kernel"}


def test_bench_prints_one_short_line_and_moves_the_rest_to_the_extras_file(tmp_path):
    """the driver could not parse round 5's 20 KB line.  The stdout line carries the contract fields, ONE roofline record and the CPU
    baseline in < 4 KB whatever the run collected; the full record goes to bench_extras.json."""
    import importlib
    import json
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    with open(os.path.join(ROOT, "profiles", "r05_final_bench_default.json")) as f:        # a real, 20 KB record of the previous round
        full = json.loads(f.read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 15000
    full["rooflines"] = full["rooflines"] * 4                                              # and it may grow
    extras = tmp_path / "bench_extras.json"
    line = bench.compact_line(full, extras_path=str(extras))
    assert "\n" not in line and len(line) < 4096
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in d, k
    assert d["config"]["workload"] and "model" not in d["config"]
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["bound"] in ("mfma", "hbm") and "traffic" in d["roofline"] and d["roofline"]["peak"]
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["kind"] in ("port", "reference")
    assert d["extras_file"] == "bench_extras.json" and json.loads(extras.read_text())["rooflines"] == full["rooflines"]
    # a sub-run (--no-extras) neither writes nor names a file
    os.remove(extras)
    d2 = json.loads(bench.compact_line(full, extras_path=str(extras), write=False))
    assert not extras.exists() and "extras_file" not in d2
    # a pathological record still yields a parseable line under the limit
    full["config"] = {k: "x" * 5000 for k in "abcdefgh"}
    full["config"]["workload"] = "w" * 5000
    assert len(bench.compact_line(full, extras_path=str(extras))) < 4096


def test_vlfuse_text_side_key_split_counts_passes_per_xcd():
    """The key split of mq_vlfuse_t2i_fwd is chosen by passes of an XCD's 32 CUs over ITS groups (group g runs on XCD g % 8): B = 8 -> 7 (7 groups
    x 9 workgroups = 63 per XCD, two passes), B = 4 -> 14 (the same 63; nsplit 7 would be 36 = two badly filled passes: 0.231 vs 0.151 ms on
    the MI355X, profiles/r04_call10_t2i_sweep.json)."""
    from mq_det_amd.modeling.pipeline import _nsplit_t2i
    assert _nsplit_t2i(8, 8, 141, 350) == 7 and _nsplit_t2i(4, 8, 141, 350) == 14
    assert _nsplit_t2i(8, 8, 141, 6) == 1                               # short key ranges are never split
    for B in (1, 2, 3, 5, 16, 31):
        ns = _nsplit_t2i(B, 8, 141, 350)
        assert 1 <= ns <= 32 and -(-350 // ns) >= 4


def test_ctypes_signatures_match_header():
    """every `_SIGNATURES` entry of mq_det_amd/ops.py has the argument kinds of its declaration in
    include/mqdet_hip.h (pointer / int / long / float, in order) -- parsed from the header text."""
    import ctypes
    import re
    from mq_det_amd import ops
    text = open(os.path.join(ROOT, "include", "mqdet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"typedef struct.*?}\s*\w+;", " ", text, flags=re.S)
    decls = dict((m.group(2), (m.group(1), m.group(3))) for m in re.finditer(r"\b(int|long)\s+(mq_\w+)\s*\(([^)]*)\)\s*;", text))
    twins = re.findall(r"^MQ_BF16_TWIN\((mq_\w+)\)", text, re.M)              # `extern decltype(name) name_bf16;`: same signature
    for n in twins:
        decls[n + "_bf16"] = decls[n]
    for n in re.findall(r"^MQ_F32_TWIN\((mq_\w+)\)", text, re.M):            # ... and name_f32 (the precise mode)
        decls[n + "_f32"] = decls[n]
    assert set(decls) == set(ops._SIGNATURES), set(decls) ^ set(ops._SIGNATURES)

    def kind(arg):
        arg = arg.strip()
        if arg in ("void", ""):
            return None
        if "*" in arg:
            return ctypes.c_void_p
        base = arg.replace("const", "").split()
        ty = " ".join(base[:-1]) if len(base) > 1 else base[0]
        return {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float}[ty]
    for name, (ret, args) in decls.items():
        want = [k for k in (kind(a) for a in args.split(",")) if k is not None]
        res, got = ops._SIGNATURES[name]
        assert got == want, (name, got, want)
        assert res == {"int": ctypes.c_int, "long": ctypes.c_long}[ret], name


def test_dcn_tile_order_round_trip_and_kv_strides():
    """ops.dcn_weight_rows inverts ops.dcn_weight_tiles (the LDS-tile order of the DCNv2 weights: exact for 16-bit dtypes, hi + lo / 2^11 for the
    split-precise planes); ops._kv_strides accepts the views the fusion layer passes to the VLFuse kernels and refuses what the kernels cannot read."""
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(5)
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        for C in (128, 256):
            w = (torch.randn(256, 9 * C, generator=g) / 48).to(dt)
            t = ops.dcn_weight_tiles(w)
            assert t.shape == w.shape and t.dtype == w.dtype and ops.dcn_is_tiled(t) and not ops.dcn_is_tiled(w)
            back = ops.dcn_weight_rows(t)
            if dt == torch.float32:
                assert float((back - w).abs().max()) <= 2.0 ** -21 * float(w.abs().max())
            else:
                assert torch.equal(back, w)
    B, T, Hh = 2, 24, 8
    pr = torch.zeros(B, T, 2 * Hh * 256 + 16, dtype=torch.float16)
    kf = pr[..., :Hh * 256].unflatten(-1, (Hh, 256)).permute(0, 2, 1, 3)
    vo = pr[..., Hh * 256:2 * Hh * 256].unflatten(-1, (Hh, 256)).permute(0, 2, 1, 3)
    assert ops._kv_strides(kf, vo) == (T * pr.shape[-1], 256, pr.shape[-1])
    assert ops._kv_strides(kf.contiguous()) == (Hh * T * 256, T * 256, 256)
    with pytest.raises(AssertionError):
        ops._kv_strides(kf, vo.contiguous())                           # keys and values must share their strides
    with pytest.raises(AssertionError):
        ops._kv_strides(pr[..., 4:Hh * 256 + 4].unflatten(-1, (Hh, 256)).permute(0, 2, 1, 3))        # rows must start on 16 bytes


def test_per_step_kernel_stats_removes_what_runs_once(tmp_path, capsys):
    """tools/per_step_kernel_stats.py: two `--stats` tables that differ only in the number of timed steps -> per-step calls / microseconds; a
    kernel that runs once per process (the model build's cast kernels) drops out, and the families add up."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import per_step_kernel_stats as ps

    def table(path, steps):
        rows = [("void dcn_igemm8_kernel<16>(DcnGroup)", 6 * steps, 500e3 * 6 * steps),
                ("Cijk_Alik_Bljk_HHS_BH", 10 * steps, 20e3 * 10 * steps),
                ("void at::native::vectorized_elementwise_kernel<4, at::native::float16_copy_kernel_cuda>", 900 + 3 * steps, 3e3 * (900 + 3 * steps)),
                ("__amd_rocclr_copyBuffer", 640, 640 * 4e3)]
        with open(path, "w") as f:
            f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
            for n, c, t in rows:
                f.write(f"\"{n}\",{c},{t},{t / c},0,0,0,0\n")
    a, b, out = tmp_path / "a.csv", tmp_path / "b.csv", tmp_path / "o.csv"
    table(a, 4)
    table(b, 24)
    sys.argv = ["per_step_kernel_stats.py", str(a), "4", str(b), "24", str(out)]
    ps.main()
    text = capsys.readouterr().out
    assert "in 19 launches" in text and "rocclr" in text
    import csv
    rows = {r["Name"]: r for r in csv.DictReader(open(out))}
    assert "__amd_rocclr_copyBuffer" not in rows                                        # ran 640 times in both runs: not part of a step
    assert float(rows["void dcn_igemm8_kernel<16>(DcnGroup)"]["CallsPerStep"]) == 6.0
    assert abs(float(rows["void dcn_igemm8_kernel<16>(DcnGroup)"]["UsPerStep"]) - 3000.0) < 1e-6
    cast = [r for n, r in rows.items() if "float16_copy" in n][0]
    assert float(cast["CallsPerStep"]) == 3.0 and abs(float(cast["UsPerStep"]) - 9.0) < 1e-6


def test_isa_loop_waits_reads_the_swin_mlp_loop():
    """tools/isa_loop_waits.py on the Swin MLP kernel: the histogram of the MFMA loops has counted LDS waits (the state of rounds 3-5 -- every wait
    lgkmcnt(0) behind a FLAT-encoded LDS copy -- would show here first) and the sequence view marks the copies."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_loop_waits as il
    txt = il.isa("swin_mlp2.hip")
    kern = dict(il.kernels(txt))
    name = [n for n in kern if n.startswith("void swin_mlp2_kernel<384, 8, true>")][0]
    nm, lg, vm = il.histogram(kern[name])
    assert nm >= 96 and sum(lg.values()) >= 40
    assert lg[0] <= sum(lg.values()) // 5, (lg[0], sum(lg.values()))                      # counted waits dominate
    seq = il.sequence(kern[name])
    assert " D " in seq and " M " in seq and "|B|" in seq


def _isa_of(src, extra=()):
    import subprocess, tempfile
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", *extra, "-S", "--cuda-device-only",
                        os.path.join(ROOT, "mq_det_amd", "csrc", src), "-o", out], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def test_lds_copies_are_buffer_loads_and_the_dcn_copy_precedes_its_gathers():
    """Static ISA checks of the kernels that stage weights by LDS-DMA (no GPU).
    (1) the copy is the MUBUF form `buffer_load_dwordx4 ... lds`: hipcc books the FLAT-encoded `global_load_lds` as a flat access and from then
        on every wait of the loop is vmcnt(0) / lgkmcnt(0) (round 6: the Swin MLP loop had no other wait than lgkmcnt(0));
    (2) the Swin MLP loop does have counted LDS waits now;
    (3) DCNv2 with LDS-tile-ordered weights: in the loop of the wave group that copies, between two barriers, every copy is issued BEFORE every
        gather and the interval ends with vmcnt(<gathers>) -- the counted wait that says "the copies have landed" is only true in that order."""
    import re
    for src, extra in (("swin_mlp2.hip", ()), ("window_attn.hip", ()), ("dcn_fused.hip", ("-fno-slp-vectorize",)),
                       ("dcn_fused.hip", ("-fno-slp-vectorize", "-DMQ_F32"))):
        txt = _isa_of(src, extra)
        assert "global_load_lds" not in txt and not re.search(r"^\s*flat_(load|store)", txt, re.M), src
        assert re.search(r"buffer_load_dwordx4 .* lds", txt), src
        if src == "swin_mlp2.hip" and not extra:
            body = re.search(r"^_Z16swin_mlp2_kernelILi384ELi8ELb1EEv14SwinMlp2Params:(.*?)s_endpgm", txt, re.S | re.M).group(1)
            counted = re.findall(r"s_waitcnt lgkmcnt\(([1-9]\d*)\)", body)
            assert len(counted) >= 8, counted
        if src == "dcn_fused.hip":
            f32 = "-DMQ_F32" in extra                     # split-precise build: 8 waves, two rows per thread
            for plain, ng in (("0", 8 if f32 else 4), ("1", 2 if f32 else 1)):
                name = (f"_ZN6mq_f3217dcn_igemm8_kernelILi8ELi0ELi1ELb{plain}ELb0ELb1EEEvNS_8DcnGroupE" if f32 else
                        f"_Z17dcn_igemm8_kernelILi16ELi0ELi1ELb{plain}ELb0ELb1EEv8DcnGroup")
                body = re.search(r"^" + name + r":(.*?)s_endpgm", txt, re.S | re.M).group(1)
                # basic blocks that hold copies AND barriers = the loop of group 1 (the fill has its copies in front of one barrier, too)
                blocks = [b for b in re.split(r"^\.LBB\d+_\d+:", body, flags=re.M) if "lds" in b and "s_barrier" in b and "v_mfma" in b]
                assert blocks, name
                for b in blocks:
                    for interval in b.split("s_barrier")[:-1]:
                        ins = [l.strip() for l in interval.split("\n")]
                        copies = [i for i, l in enumerate(ins) if re.match(r"buffer_load_dwordx4 .* lds", l)]
                        gathers = [i for i, l in enumerate(ins) if l.startswith("global_load_dwordx4")]
                        if not copies:
                            continue
                        assert gathers and max(copies) < min(gathers), (name, "a gather is issued before the last copy")
                        assert len(gathers) == ng, (name, len(gathers))
                        waits = [l for l in ins[max(gathers):] if l.startswith("s_waitcnt") and "vmcnt" in l]
                        assert waits and re.search(r"vmcnt\((\d+)\)", waits[-1]).group(1) in (str(ng), "0"), (name, waits)
                        assert any(f"vmcnt({ng})" in w for w in waits), (name, waits)


def test_gather_kernels_keep_their_loads_in_flight():
    """Static ISA check (tools/isa_wait_scan.py, no GPU): the MSDeformAttn and window-attention kernels must not fall back to
    one `s_waitcnt vmcnt(0)` per global load -- the pattern that made the first MSDeformAttn kernel 2.3x slower."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_wait_scan
    for f in ("msda.hip", "window_attn.hip"):
        rows = isa_wait_scan.scan(os.path.join(ROOT, "mq_det_amd", "csrc", f))
        assert rows, f
        for name, loads, waits, immediate in rows:
            # window_attn_qkv_kernel: its two one-time global -> LDS staging copies (bias, weights) are load / wait / store by nature, and
            # one spilled register is reloaded behind the X loads at C = 96 (13 dwords of scratch at the 256-VGPR cap)
            allowed = 3 if "window_attn_qkv" in name else 0
            assert immediate <= allowed, f"{f}: {name}: {immediate} of {loads} loads are waited on immediately"
