"""CPU tests of host-side edge cases on the hot path: ragged / empty label maps, banks with fewer queries than asked,
ragged batches, empty detections, sharding arithmetic.  No GPU, no HIP calls (the oracle is the checker)."""
import torch

from mq_det_amd import get_cfg, parallel
from mq_det_amd.modeling import pipeline
from mq_det_amd.modeling.query_selector import QuerySelector, build_token_index, labels_and_maps
from oracle import detector as od
from oracle import postprocess as op


def test_token_index_ragged_and_empty_labels():
    pm = {3: [1, 2, 5], 7: [9], 11: [], 12: [20, 21]}
    labels = [k for k, v in pm.items() if len(v)]
    idx, ids = build_token_index(pm, labels, torch.device("cpu"))
    assert ids.tolist() == [3, 7, 12]                       # label 11 has no token -> dropped like the reference
    assert idx.shape == (3, 3) and idx.dtype == torch.int32
    assert idx.tolist() == [[1, 2, 5], [9, -1, -1], [20, 21, -1]]
    idx0, ids0 = build_token_index({}, [], torch.device("cpu"))   # caption without any positive: well-formed, all padding
    assert idx0.shape == (1, 1) and int(idx0[0, 0]) == -1 and ids0.numel() == 1


def test_labels_and_maps_normalisation():
    pm = {1: [2, 3], 2: [], 5: [7]}
    labels, m = labels_and_maps(pm, 12)
    assert labels == [1, 5] and m.shape == (2, 12)
    assert torch.allclose(m.sum(1), torch.ones(2), atol=1e-5)
    ref_labels, ref_m = od.labels_and_maps(pm, 12)
    assert labels == ref_labels and torch.allclose(m, ref_m)


def test_query_selector_short_bank_and_ragged_batch():
    """A label with fewer queries than NUM_QUERY_PER_CLASS contributes what it has; images of one batch may carry
    different label sets (zero-padded vision rows, -1 padded gather index); index == nonzeros of the reference mask."""
    cfg = get_cfg()
    C, k, T = cfg.MODEL.BACKBONE.OUT_CHANNELS, cfg.VISION_QUERY.NUM_QUERY_PER_CLASS, 32
    g = torch.Generator().manual_seed(3)
    bank = {1: torch.randn(k, 1, C, generator=g), 2: torch.randn(2, 1, C, generator=g), 4: torch.randn(k + 3, 1, C, generator=g)}
    pm_a = {1: [1, 2], 2: [4], 4: [6, 7, 8]}
    pm_b = {4: [3]}
    qs = QuerySelector(cfg)
    qs.load_query_bank(bank)
    la, lb = list(pm_a), list(pm_b)
    # label 4 holds more rows than k: like the reference (query_selector.py:74-76) the rows are drawn with
    # sorted(np.random.choice(len, k, replace=False)) from numpy's global generator -> same seed, same draw
    import numpy as np
    np.random.seed(11)
    vision, idx = qs.select([la, lb], [pm_a, pm_b], T, torch.device("cpu"), torch.float32)
    _, amap_a = od.labels_and_maps(pm_a, T)
    _, amap_b = od.labels_and_maps(pm_b, T)
    np.random.seed(11)
    ref_v, ref_m = od.select_queries(bank, [la, lb], [amap_a, amap_b], k)
    assert not qs.deterministic(la) and qs.deterministic([1, 2])       # only deterministic selections are memoised
    assert vision.shape == ref_v.shape == (2, k + 2 + k, C)
    assert torch.equal(vision, ref_v)
    for b in range(2):
        for t in range(T):
            want = torch.nonzero(ref_m[b, :, t]).flatten().tolist()
            got = [i for i in idx[b, t].tolist() if i >= 0]
            assert got == want, (b, t, got, want)
    assert idx.shape[2] == k                                  # widest token owns one label's rows (label 4 capped at k)


def test_query_selector_reference_rng_stream():
    """with VISION_QUERY.REFERENCE_RNG_STREAM every label of every forward consumes numpy's global generator like the
    reference (query_selector.py:74), identity draws included: after a caption that needed NO real draw the generator is where the
    reference's is, so a later caption that does need one selects the same rows."""
    import numpy as np
    cfg = get_cfg()
    cfg.VISION_QUERY.REFERENCE_RNG_STREAM = True
    C, k, T = cfg.MODEL.BACKBONE.OUT_CHANNELS, cfg.VISION_QUERY.NUM_QUERY_PER_CLASS, 16
    g = torch.Generator().manual_seed(8)
    bank = {1: torch.randn(k, 1, C, generator=g), 2: torch.randn(2, 1, C, generator=g), 3: torch.randn(k + 4, 1, C, generator=g)}
    pm_easy, pm_hard = {1: [1], 2: [3]}, {3: [2, 4], 1: [6]}
    qs = QuerySelector(cfg)
    qs.load_query_bank(bank)
    np.random.seed(21)
    qs.select_cached("easy", list(pm_easy), pm_easy, 1, T, torch.device("cpu"), torch.float32)        # identity draws only
    qs.select_cached("easy", list(pm_easy), pm_easy, 1, T, torch.device("cpu"), torch.float32)        # ... and NOT memoised
    v, _ = qs.select_cached("hard", list(pm_hard), pm_hard, 1, T, torch.device("cpu"), torch.float32)
    np.random.seed(21)
    for pm in (pm_easy, pm_easy):
        od.select_queries(bank, [list(pm)], [od.labels_and_maps(pm, T)[1]], k)
    ref_v, _ = od.select_queries(bank, [list(pm_hard)], [od.labels_and_maps(pm_hard, T)[1]], k)
    assert torch.equal(v, ref_v)
    cfg.VISION_QUERY.REFERENCE_RNG_STREAM = False                        # default: identity draws are skipped -> the state differs
    qs2 = QuerySelector(cfg)
    qs2.load_query_bank(bank)
    assert qs2.deterministic(list(pm_easy)) and not qs.deterministic(list(pm_easy))


def test_query_selector_defaultdict_bank_with_empty_labels():
    """reference banks are `defaultdict(list)` (engine/inference.py:401); a caption label without queries reads
    as `[]` and contributes no vision rows (query_selector.py:77-78) -- text-only for that label, no exception; a plain
    dict that lacks the key behaves the same."""
    from collections import defaultdict
    cfg = get_cfg()
    C, k, T = cfg.MODEL.BACKBONE.OUT_CHANNELS, cfg.VISION_QUERY.NUM_QUERY_PER_CLASS, 16
    g = torch.Generator().manual_seed(4)
    rows = {2: torch.randn(k, 1, C, generator=g), 7: torch.randn(3, 1, C, generator=g)}
    pm = {1: [1], 2: [3, 4], 5: [6], 7: [8]}
    _, amap = od.labels_and_maps(pm, T)
    for bank in (defaultdict(list, rows), dict(rows)):
        qs = QuerySelector(cfg)
        qs.load_query_bank(bank)
        vision, idx = qs.select([list(pm)], [pm], T, torch.device("cpu"), torch.float32)
        assert vision.shape == (1, k + 3, C)
        assert torch.equal(vision[0, :k], rows[2].flatten(0, 1)) and torch.equal(vision[0, k:], rows[7].flatten(0, 1))
        assert [i for i in idx[0, 1].tolist() if i >= 0] == [] and [i for i in idx[0, 6].tolist() if i >= 0] == []
        assert [i for i in idx[0, 3].tolist() if i >= 0] == list(range(k)) and [i for i in idx[0, 8].tolist() if i >= 0] == [k, k + 1, k + 2]
        q, m, has = qs([list(pm)], [amap])
        assert torch.equal(q, vision) and has == [[0, 1, 0, 1]]
        if isinstance(bank, defaultdict):                      # the oracle follows the reference on the same bank
            ref_v, ref_m = od.select_queries(bank, [list(pm)], [amap], k)
            assert torch.equal(ref_v, vision) and torch.equal(ref_m, m)
    # a caption none of whose labels has a query -> zero vision rows (the detector then runs text-only)
    qs = QuerySelector(cfg)
    qs.load_query_bank(defaultdict(list, rows))
    v0, i0 = qs.select([[1, 5]], [{1: [1], 5: [2]}], T, torch.device("cpu"), torch.float32)
    assert v0.shape == (1, 0, C) and int((i0 >= 0).sum()) == 0


def test_pyramid_token_views_round_trip():
    g = torch.Generator().manual_seed(5)
    sizes = [(6, 7), (3, 4), (2, 2)]
    feats = [torch.randn(2, 16, h, w, generator=g).contiguous(memory_format=torch.channels_last) for h, w in sizes]
    tok, got_sizes = pipeline._to_tokens(feats)
    tok = tok.contiguous()
    assert got_sizes == sizes and tok.shape == (2, sum(h * w for h, w in sizes), 16)
    views = pipeline._level_views(tok, sizes)
    for v, f in zip(views, feats):
        assert v.shape == f.shape and torch.equal(v, f)
    views[1][:] = 0                                          # views alias the token buffer (no copies)
    off = sizes[0][0] * sizes[0][1]
    assert float(tok[:, off:off + sizes[1][0] * sizes[1][1]].abs().sum()) == 0.0
    assert torch.equal(pipeline._level_views(tok, sizes)[0], feats[0])


def test_nsplit_heuristic_bounds():
    assert pipeline._nsplit(1024, 400) == 1                  # enough workgroups already
    assert pipeline._nsplit(64, 4) == 1                      # too few key tiles to split
    for blocks, tiles in ((64, 350), (16, 88), (128, 350), (1, 1000)):
        n = pipeline._nsplit(blocks, tiles)
        assert 1 <= n <= 32 and n <= max(1, tiles // 4)


def test_shard_range_and_detection_packing():
    assert parallel.shard_range(10, 0, 4) == [0, 1, 2] and parallel.shard_range(10, 3, 4) == [9, 0, 1]   # wrap-around tail
    assert sorted(set(sum((parallel.shard_range(10, r, 4) for r in range(4)), []))) == list(range(10))
    boxes = torch.zeros(2, 3, 4)
    boxes[0, 0] = torch.tensor([1.0, 2.0, 3.0, 4.0])
    scores = torch.tensor([[0.9, -1.0, -1.0], [-1.0, -1.0, -1.0]])          # image 1: no detection at all
    labels = torch.tensor([[5, 0, 0], [0, 0, 0]])
    packed = parallel.pack_detections(boxes, scores, labels)
    assert packed.shape == (2, 3, 6) and parallel.gather_detections(packed) is packed          # world size 1: identity
    out = parallel.unpack_detections(packed)
    assert len(out[0]["boxes"]) == 1 and out[0]["labels"].tolist() == [5] and len(out[1]["boxes"]) == 0


def test_oracle_nms_degenerate_inputs():
    """Class-aware NMS of 0 / 1 / duplicate boxes (reference csrc/cuda/ml_nms.cu:15-26: IoU 0 across labels)."""
    assert op.ml_nms(torch.zeros(0, 4), torch.zeros(0), torch.zeros(0, dtype=torch.long), 0.6).numel() == 0
    assert op.ml_nms(torch.tensor([[0.0, 0.0, 5.0, 5.0]]), torch.tensor([0.3]), torch.tensor([1]), 0.6).tolist() == [0]
    # identical boxes, different labels: class-aware NMS keeps both; same label: keeps the higher score only
    b = torch.tensor([[0.0, 0.0, 9.0, 9.0], [0.0, 0.0, 9.0, 9.0]])
    assert sorted(op.ml_nms(b, torch.tensor([0.5, 0.6]), torch.tensor([1, 2]), 0.6).tolist()) == [0, 1]
    assert op.ml_nms(b, torch.tensor([0.5, 0.6]), torch.tensor([2, 2]), 0.6).tolist() == [1]


def test_groundingdino_host_glue_edge_cases(tmp_path):
    """MQ-GroundingDINO host side without a GPU: config validation by key name, class-score map (mean over a label's tokens,
    empty label flagged, token positions beyond max_text_len dropped by forward), caption pre-processing, geometry of a batch
    without padding and of one whose images are smaller than the padded canvas."""
    import pytest
    import torch
    from mq_det_amd.config import get_gdino_cfg
    from mq_det_amd.modeling import gdino, gdino_pipeline as gp
    from mq_det_amd.utils.tokenizer import build_synthetic_tokenizer
    cfg = get_gdino_cfg()
    cfg.GROUNDINGDINO.text_encoder_type = build_synthetic_tokenizer(str(tmp_path), size=2048)
    cfg.MODEL.LANGUAGE_BACKBONE.BERT_VOCAB_SIZE = 2048
    cfg.GROUNDINGDINO.swin_depths, cfg.GROUNDINGDINO.enc_layers, cfg.GROUNDINGDINO.dec_layers = (2, 2, 2, 2), 1, 1
    cfg.MODEL.LANGUAGE_BACKBONE.NUM_HIDDEN_LAYERS = 7
    model = gdino.GroundingDINO(cfg)
    assert gdino.preprocess_caption("  Cat. Dog ") == "cat. dog." and gdino.preprocess_caption("a.") == "a."
    cmap, empty = model._class_map({1: [1, 2], 3: [5], 4: []}, torch.device("cpu"))
    assert empty and cmap.shape == (256, 80) and float(cmap[1, 0]) == 0.5 and float(cmap[5, 2]) == 1.0 and float(cmap[:, 3].sum()) == 0.0
    cmap, empty = model._class_map({2: 7}, torch.device("cpu"))                  # a single token given as an int
    assert not empty and float(cmap[7, 1]) == 1.0
    with pytest.raises(RuntimeError, match="MI355X only"):
        model.prepare(torch.device("cpu"))                                      # no CPU fallback
    bad = get_gdino_cfg()
    bad.GROUNDINGDINO.text_encoder_type = cfg.GROUNDINGDINO.text_encoder_type
    bad.GROUNDINGDINO.two_stage_type = "no"
    with pytest.raises(NotImplementedError, match="two_stage_type"):
        gdino.GroundingDINO(bad)
    with pytest.raises(NotImplementedError, match="inference forward only"):
        model.train()
    # geometry: no padding at all -> no key mask work; padded -> valid extents per level
    P = {"level_embed32": torch.zeros(4, 256), "transformer.level_embed": torch.zeros(4, 256)}
    g0 = gp.geometry(P, cfg, 128, 160, [(128, 160)], torch.device("cpu"))
    assert not g0["any_pad"] and g0["shapes"] == ((16, 20), (8, 10), (4, 5), (2, 3)) and bool((g0["vr"] == 1).all())
    g1 = gp.geometry(P, cfg, 128, 160, [(100, 160), (128, 90)], torch.device("cpu"))
    assert g1["any_pad"] and g1["valid_hw"].tolist()[0][0] == [13, 20] and g1["valid_hw"].tolist()[1][0] == [16, 12]
    assert torch.isinf(g1["proposals"][0][g1["mask"][0]]).all() and int(g1["key_mask"].shape[1]) % 64 == 0
    assert gp.level_shapes(800, 1344, 4) == [(100, 168), (50, 84), (25, 42), (13, 21)]
