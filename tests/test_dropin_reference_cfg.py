"""Drop-in evidence: the product model is built from the reference's REAL config tree
(`maskrcnn_benchmark/config/defaults.py` + `configs/pretrain/mq-glip-t.yaml` + `configs/vision_query_5shot/lvis_minival.yaml`,
loaded in place through oracle/_refload.py) and driven through the reference's own caller sequence:

  * captions / positive maps come from the reference's `create_queries_and_maps_from_dataset` (engine/inference.py:195-283,
    executed in place) on an LVIS-like category list, chunked by TEST.CHUNKED_EVALUATION = 40;
  * the loop body is engine/inference.py:599-643: `images.to(device)`, one `model(images, captions=, positive_map=)` per chunk,
    `[o.to(cpu_device) for o in output]`, `resize_box`, `.extra_fields["scores"|"labels"]`, `.bbox` -- the LVIS branch;
  * tools/test_grounding_net.py:141-154: `build_detection_model(cfg)`, `model.to(cfg.MODEL.DEVICE)` (skipped: no GPU here),
    checkpoint `load_state_dict`, `model.load_query_bank`.

The box has no GPU, so the HIP entry points are swapped for the test-only torch emulations (tests/ops_emulation.py) -- what is
under test here is the BOUNDARY: every cfg attribute the product reads exists in the reference tree, the call signatures,
return types and BoxList methods the callers use, and (against the oracle on the same weights) the results.
Needs /root/reference (build container only)."""
import os
import sys
import tempfile
import types

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout (build container)")

LVIS_LIKE = ["aerosol_can", "air_conditioner", "airplane", "alarm_clock", "alcohol", "alligator", "almond", "ambulance", "amplifier",
             "anklet", "antenna", "apple", "applesauce", "apricot", "apron", "aquarium", "arctic_(type_of_shoe)", "armband", "armchair",
             "armoire", "armor", "artichoke", "trash_can", "ashtray", "asparagus", "atomizer", "avocado", "award", "awning", "ax",
             "baboon", "baby_buggy", "basketball_backboard", "backpack", "handbag", "suitcase", "bagel", "bagpipe", "baguet", "bait",
             "ball", "ballet_skirt", "balloon", "bamboo", "banana", "Band_Aid", "bandage", "bandanna", "banjo", "banner"]


def test_reference_cfg_tree_and_caller_sequence(monkeypatch):
    import ops_emulation as emu
    from oracle import _refload, detector as od
    from oracle.spec import Spec
    from oracle.weights import make_state_dict, make_query_bank
    import mq_det_amd
    from mq_det_amd import ops
    from mq_det_amd.modeling import pipeline, detector
    from mq_det_amd.structures import to_image_list
    from mq_det_amd.utils.tokenizer import build_synthetic_tokenizer

    # ---- the reference's config tree, exactly as tools/test_grounding_net.py:100-106 builds it (cfg.merge_from_file x 2)
    cfg = _refload.reference_cfg("configs/pretrain/mq-glip-t.yaml", "configs/vision_query_5shot/lvis_minival.yaml")
    assert cfg.MODEL.META_ARCHITECTURE == "GeneralizedVLRCNN_New" and cfg.TEST.CHUNKED_EVALUATION == 40
    assert cfg.TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM == 3000 and cfg.MODEL.ATSS.DETECTIONS_PER_IMG == 300
    # command-line style overrides (cfg.merge_from_list in the reference): a shallow model so that the CPU emulation is
    # quick, a local tokenizer directory (no network), no bank file (loaded through load_query_bank below)
    words = [w for n in LVIS_LIKE for w in n.lower().replace("(", " ").replace(")", " ").replace("_", " ").split()]
    tok_dir = build_synthetic_tokenizer(tempfile.mkdtemp(), size=4000, extra_words=words)
    cfg.MODEL.SWINT.DEPTHS = (2, 2, 2, 2)
    cfg.MODEL.DYHEAD.NUM_CONVS = 2
    cfg.MODEL.LANGUAGE_BACKBONE.TOKENIZER_TYPE = tok_dir
    cfg.MODEL.LANGUAGE_BACKBONE.MODEL_TYPE = tok_dir
    cfg.VISION_QUERY.QUERY_BANK_PATH = ""
    spec = Spec(swin_depths=(2, 2, 2, 2), dyhead_convs=2, vocab=4000, num_classes=1204)
    # the BERT depth is not in the reference's config tree (HF bert-base-uncased has 12 layers, 6 GCP blocks): keep it
    sd = make_state_dict(spec, 0)

    # ---- test-only: emulated HIP entry points, CPU plan (the product refuses CPU by design)
    emu.patch_into(monkeypatch, ops)

    def prepare(self, device=None):
        self._plan = pipeline.build_plan(self.state_dict(), self.cfg, torch.device("cpu"), dtype=torch.float32)
        self._plan_key = torch.device("cpu")
        self.use_hip_graph = False
        return self._plan
    monkeypatch.setattr(detector.GeneralizedVLRCNN_New, "prepare", prepare)
    cfg.MODEL.DYHEAD.LEVEL_STREAMS = False                    # product-only key (side streams need a GPU)

    # ---- tools/test_grounding_net.py:141-154
    model = mq_det_amd.build_detection_model(cfg)
    assert type(model).__name__ == cfg.MODEL.META_ARCHITECTURE
    model.load_state_dict(sd, strict=True)                     # DetectronCheckpointer.load ends in a strict load
    model.eval()
    model._validate_config()                                   # every key it reads exists in the reference tree; mq-glip-t passes

    # ---- engine/inference.py: captions and positive maps from the reference's own builders
    fns = _refload.reference_functions("maskrcnn_benchmark/engine/inference.py",
                                       ["clean_name", "create_positive_dict", "chunks", "create_queries_and_maps",
                                        "create_queries_and_maps_from_dataset", "resize_box"],
                                       {"load_from_yaml_file": None})
    dataset = types.SimpleNamespace(categories=lambda: {i + 1: n for i, n in enumerate(LVIS_LIKE)})
    all_queries, all_maps = fns["create_queries_and_maps_from_dataset"](dataset, cfg, disable_print=True)
    assert len(all_queries) == 2 and all_queries[0].startswith("aerosol can. air conditioner. airplane")
    assert sorted(all_maps[0]) == list(range(1, 41)) and sorted(all_maps[1]) == list(range(41, 51))
    bank = make_query_bank(range(1, 51), spec)
    model.load_query_bank(bank)                                # engine/inference.py:411

    # ---- the eval loop body, engine/inference.py:599-643 (LVIS branch), one image per batch like TEST.IMS_PER_BATCH = 1
    g = torch.Generator().manual_seed(0)
    img = torch.randn(3, 150, 190, generator=g)
    images = to_image_list([img], cfg.DATALOADER.SIZE_DIVISIBILITY)
    targets = [{"orig_size": torch.tensor([300, 380]), "image_id": torch.tensor(7)}]
    device, cpu_device = torch.device("cpu"), torch.device("cpu")
    mdetr_style_output = []
    with torch.no_grad():
        images = images.to(device)
        query_time = len(all_queries)
        for query_i in range(query_time):
            captions = [all_queries[query_i] for ii in range(len(targets))]
            positive_map_label_to_token = all_maps[query_i]
            output = model(images, captions=captions, positive_map=positive_map_label_to_token)
            output = [o.to(cpu_device) for o in output]
            output = output[0]
            output = fns["resize_box"](output, targets)
            scores = output.extra_fields["scores"]
            labels = output.extra_fields["labels"]
            boxes = output.bbox
            mdetr_style_output.append((targets[0]["image_id"].item(), {"scores": scores, "labels": labels, "boxes": boxes}))
    assert model.cache_stats["backbone_miss"] == 1 and model.cache_stats["backbone_hit"] == 1     # same pixels, second chunk
    for (_, o), labs in zip(mdetr_style_output, (range(1, 41), range(41, 51))):
        assert o["scores"].dtype == torch.float32 and o["labels"].dtype == torch.int64 and o["boxes"].shape[1] == 4
        assert len(o["scores"]) <= cfg.MODEL.ATSS.DETECTIONS_PER_IMG and len(o["scores"]) > 0
        assert set(o["labels"].tolist()) <= set(labs)
        assert float(o["boxes"][:, 2].max()) <= 380 and float(o["boxes"][:, 3].max()) <= 300      # resized to orig_size

    # ---- chunk batching: all chunk captions of the image in ONE call == the per-chunk calls above
    with torch.no_grad():
        batched = model.forward_chunks(images, list(zip(all_queries, all_maps)))
    assert len(batched) == len(all_queries) and all(len(r) == 1 for r in batched)
    for res, (_, o) in zip(batched, mdetr_style_output):
        r = fns["resize_box"](res[0].to(cpu_device), targets)
        o1, o2 = torch.argsort(o["scores"], descending=True, stable=True), torch.argsort(r.extra_fields["scores"], descending=True, stable=True)
        assert len(o["scores"]) == len(r.extra_fields["scores"])
        assert torch.allclose(o["scores"][o1], r.extra_fields["scores"][o2], atol=1e-5)
        assert torch.equal(o["labels"][o1], r.extra_fields["labels"][o2])
        assert torch.allclose(o["boxes"][o1], r.bbox[o2], atol=1e-3)

    # ---- same weights, same inputs through the oracle: the detections of chunk 0 agree (fp32 emulation: to rounding)
    tk = model.tokenizer
    t = tk([all_queries[0]], max_length=256, padding="max_length", return_special_tokens_mask=True, return_tensors="pt", truncation=True)
    pimg, sizes = od.pad_images([img], 32)
    spec_pm = dict(all_maps[0])
    dets = od.forward(sd, spec, pimg, sizes, t["input_ids"], t["attention_mask"], spec_pm, bank)
    ref, got = dets[0], mdetr_style_output[0][1]
    assert len(ref["scores"]) == len(got["scores"])
    o1, o2 = torch.argsort(ref["scores"], descending=True, stable=True), torch.argsort(got["scores"], descending=True, stable=True)
    assert torch.allclose(ref["scores"][o1], got["scores"][o2], atol=2e-4)
    assert torch.equal(ref["labels"][o1], got["labels"][o2])
    assert torch.allclose(ref["boxes"][o1] * 2.0, got["boxes"][o2], atol=2e-2)                    # resize_box: 150x190 -> 300x380


def test_groundingdino_reference_cfg_and_caller_sequence(monkeypatch):
    """The same evidence for the MQ-GroundingDINO family (BASELINE configs[4]): reference config tree = defaults.py +
    configs/pretrain/mq-groundingdino-t.yaml; `build_detection_model(cfg)` switches on `cfg.GROUNDINGDINO.enabled`
    (modeling/detector/__init__.py:9-11); strict load of a state_dict with the reference module's names; the eval loop body of
    engine/inference.py:599-643 on captions / positive maps from the reference's own builders; results vs the oracle (which is
    pinned to the reference's GroundingDINO module by tests/test_oracle_golden.py)."""
    import ops_emulation as emu
    from oracle import _refload, gdino as og
    from oracle.spec import tiny_gdino_spec
    from oracle.weights import make_gdino_state_dict, make_query_bank
    import mq_det_amd
    from mq_det_amd import ops
    from mq_det_amd.modeling import gdino, gdino_pipeline as gp
    from mq_det_amd.structures import to_image_list
    from mq_det_amd.utils.tokenizer import build_synthetic_tokenizer

    cfg = _refload.reference_cfg("configs/pretrain/mq-groundingdino-t.yaml")
    assert cfg.GROUNDINGDINO.enabled and cfg.GROUNDINGDINO.num_queries == 900 and cfg.GROUNDINGDINO.box_threshold == 0.05
    names = LVIS_LIKE[:12]
    words = [w for n in names for w in n.lower().replace("(", " ").replace(")", " ").replace("_", " ").split()]
    spec = tiny_gdino_spec(vocab=4000)
    tok_dir = build_synthetic_tokenizer(tempfile.mkdtemp(), size=spec.vocab, extra_words=words)
    # command-line style overrides: shallow model (product-only keys for the depths the reference hard-codes by name), local tokenizer
    G = cfg.GROUNDINGDINO
    G.enc_layers, G.dec_layers, G.num_queries, G.text_encoder_type = spec.enc_layers, spec.dec_layers, spec.num_queries, tok_dir
    G.swin_depths = spec.swin_depths
    cfg.MODEL.LANGUAGE_BACKBONE.TOKENIZER_TYPE = cfg.MODEL.LANGUAGE_BACKBONE.MODEL_TYPE = tok_dir     # the caption builders read it
    cfg.MODEL.LANGUAGE_BACKBONE.NUM_HIDDEN_LAYERS, cfg.MODEL.LANGUAGE_BACKBONE.QV_START = spec.bert_layers, spec.qv_start
    cfg.MODEL.LANGUAGE_BACKBONE.BERT_VOCAB_SIZE = spec.vocab
    cfg.VISION_QUERY.QUERY_BANK_PATH = ""
    sd = make_gdino_state_dict(spec, 0)

    emu.patch_into(monkeypatch, ops)

    def prepare(self, device=None):
        self._plan = gp.build_gdino_plan(self.state_dict(), self.cfg, torch.device("cpu"), self._swin, dtype=torch.float32)
        self._plan_key = torch.device("cpu")
        self.use_hip_graph = False
        return self._plan
    monkeypatch.setattr(gdino.GroundingDINO, "prepare", prepare)

    model = mq_det_amd.build_detection_model(cfg)              # tools/test_grounding_net.py:141
    assert type(model).__name__ == "GroundingDINO"
    model.load_state_dict(sd, strict=True)
    model.eval()

    fns = _refload.reference_functions("maskrcnn_benchmark/engine/inference.py",
                                       ["clean_name", "create_positive_dict", "chunks", "create_queries_and_maps",
                                        "create_queries_and_maps_from_dataset", "resize_box"],
                                       {"load_from_yaml_file": None})
    dataset = types.SimpleNamespace(categories=lambda: {i + 1: n for i, n in enumerate(names)})
    all_queries, all_maps = fns["create_queries_and_maps_from_dataset"](dataset, cfg, disable_print=True)
    assert len(all_queries) == 1 and sorted(all_maps[0]) == list(range(1, 13))
    bank = make_query_bank(range(1, 13), spec, seed=1, scales=1)
    model.load_query_bank(bank)

    g = torch.Generator().manual_seed(0)
    img = torch.randn(3, 120, 150, generator=g)
    images = to_image_list([img], cfg.DATALOADER.SIZE_DIVISIBILITY)
    targets = [{"orig_size": torch.tensor([240, 300]), "image_id": torch.tensor(3)}]
    with torch.no_grad():
        output = model(images, captions=[all_queries[0]], positive_map=all_maps[0])
        output = [o.to(torch.device("cpu")) for o in output][0]
        res, feats = model(images, captions=[all_queries[0]], positive_map=all_maps[0], return_backbone_features=True)
        # second call with the same pixels: Swin + input projections come from the per-image cache, same result
        assert model.cache_stats["backbone_miss"] == 1 and model.cache_stats["backbone_hit"] == 1
        assert len(res[0]) == len(output) and torch.equal(res[0].bbox, output.bbox)
        assert torch.equal(res[0].get_field("scores"), output.get_field("scores"))
        images.tensors.add_(0.0)                                   # an in-place write bumps the version counter: cache miss
        model(images, captions=[all_queries[0]], positive_map=all_maps[0])
        assert model.cache_stats["backbone_miss"] == 2
        output = fns["resize_box"](output, targets)
    scores, labels, boxes = output.extra_fields["scores"], output.extra_fields["labels"], output.bbox
    assert scores.dtype == torch.float32 and labels.dtype == torch.int64 and 0 < len(scores) <= spec.num_queries
    assert set(labels.tolist()) <= set(range(1, 13)) and float(scores.min()) > cfg.GROUNDINGDINO.box_threshold
    assert len(feats) == 4 and feats[0].shape[1] == 256 and feats[0].shape[-2:] == (16, 20)      # `srcs` for online_update

    # the oracle on the same weights / pixels / caption (preprocess_caption adds the final '.')
    tok = model.tokenizer([gdino.preprocess_caption(all_queries[0])], padding="max_length", return_tensors="pt")
    with torch.no_grad():
        o = og.forward(sd, spec, images.tensors, [tuple(s) for s in images.image_sizes], tok["input_ids"], tok["attention_mask"],
                       dict(all_maps[0]), model.specical_tokens, bank)
    bx, sc, lb = o["detections"][0]
    assert len(bx) == len(scores)
    assert torch.allclose(sc, scores, atol=2e-4) and torch.equal(lb, labels)
    assert torch.allclose(bx * 2.0, boxes, atol=2e-2)                                             # resize_box: 120x150 -> 240x300

    # extract_query on the projected levels (groundingdino.py:340-421) appends pooled features to the bank
    from mq_det_amd.structures import BoxList
    t = BoxList(torch.tensor([[10.0, 12.0, 80.0, 90.0], [30.0, 20.0, 140.0, 110.0]]), (150, 120), mode="xyxy")
    t.add_field("labels", torch.tensor([2, 5]))
    from collections import defaultdict
    qi = model.extract_query(samples=images, targets=[t], query_images=defaultdict(list), visual_features=feats)
    assert sorted(qi) == [2, 5] and qi[2].shape == (1, 1, 256)


def test_glipdemo_caller_sequence(monkeypatch):
    """north_star: "keeping the GeneralizedVLRCNN / GLIPDemo API surface".  The reference's `GLIPDemo.compute_prediction`,
    `_post_process_fixed_thresh` and `_post_process` (engine/predictor_glip.py:178-275) and its positive-map builders (:405-445)
    are executed IN PLACE on the product model: category-list caption (" . " separators), `create_positive_map` from the
    tokenizer's char_to_token, `plus = 1` label ids, `model(image_list, captions=[caption], positive_map=...)` with the default
    (non-LVIS) score aggregation, `prediction.resize`, `has_field("mask")`, `predictions[keep]`, `.sort` -- results vs the oracle."""
    import timeit
    import numpy as np
    import ops_emulation as emu
    from transformers import AutoTokenizer
    from oracle import _refload, detector as od
    from oracle.spec import Spec
    from oracle.weights import make_state_dict
    import mq_det_amd
    from mq_det_amd import ops
    from mq_det_amd.modeling import pipeline, detector
    from mq_det_amd.structures import to_image_list
    from mq_det_amd.utils.tokenizer import build_synthetic_tokenizer

    names = ["person", "sports ball", "traffic light", "dog"]
    cfg = _refload.reference_cfg("configs/pretrain/mq-glip-t.yaml")
    assert cfg.TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM == -1 and cfg.MODEL.RPN_ARCHITECTURE == "VLDYHEAD"
    tok_dir = build_synthetic_tokenizer(tempfile.mkdtemp(), size=4000, extra_words=[w for n in names for w in n.split()])
    cfg.MODEL.SWINT.DEPTHS, cfg.MODEL.DYHEAD.NUM_CONVS = (2, 2, 2, 2), 2
    cfg.MODEL.LANGUAGE_BACKBONE.TOKENIZER_TYPE = cfg.MODEL.LANGUAGE_BACKBONE.MODEL_TYPE = tok_dir
    cfg.VISION_QUERY.QUERY_BANK_PATH = ""
    cfg.MODEL.DYHEAD.LEVEL_STREAMS = False
    spec = Spec(swin_depths=(2, 2, 2, 2), dyhead_convs=2, vocab=4000, num_classes=cfg.MODEL.DYHEAD.NUM_CLASSES, mdetr_class_num=-1,
                detections_per_img=cfg.MODEL.ATSS.DETECTIONS_PER_IMG, vision_query=False)
    sd = make_state_dict(Spec(swin_depths=(2, 2, 2, 2), dyhead_convs=2, vocab=4000, num_classes=cfg.MODEL.DYHEAD.NUM_CLASSES), 0)
    emu.patch_into(monkeypatch, ops)

    def prepare(self, device=None):
        self._plan = pipeline.build_plan(self.state_dict(), self.cfg, torch.device("cpu"), dtype=torch.float32)
        self._plan_key = torch.device("cpu")
        self.use_hip_graph = False
        return self._plan
    monkeypatch.setattr(detector.GeneralizedVLRCNN_New, "prepare", prepare)

    ref = _refload.reference_classes("maskrcnn_benchmark/engine/predictor_glip.py", ["GLIPDemo"],
                                     ["create_positive_map", "create_positive_map_label_to_token_from_positive_map"],
                                     {"to_image_list": to_image_list, "timeit": timeit, "print": lambda *a, **k: None})
    demo = ref["GLIPDemo"].__new__(ref["GLIPDemo"])            # GLIPDemo.__init__ minus checkpoint / cv2 / nltk plumbing (:29-61)
    demo.cfg = cfg
    demo.model = mq_det_amd.build_detection_model(cfg)
    demo.model.load_state_dict(sd, strict=True)
    demo.model.eval()
    demo.device = demo.cpu_device = torch.device("cpu")
    demo.tokenizer = AutoTokenizer.from_pretrained(tok_dir)     # build_tokenizer (:93-106)
    mean, std = torch.tensor(cfg.INPUT.PIXEL_MEAN).view(3, 1, 1), torch.tensor(cfg.INPUT.PIXEL_STD).view(3, 1, 1)
    demo.transforms = lambda img: (torch.from_numpy(img).permute(2, 0, 1).float() - mean) / std     # build_transform without the resize
    demo.confidence_threshold = 0.06
    g = np.random.default_rng(0)
    image = (g.random((150, 190, 3)) * 255).astype(np.float32)

    # (a) a list of category names: the reference joins them with " . " and wraps ALL spans into ONE entity (:185-196), i.e. a
    # single label whose score is the mean over every token of the caption
    pred = demo.compute_prediction(image, names)
    assert demo.plus == 1 and sorted(demo.positive_map_label_to_token) == [1] and len(demo.positive_map_label_to_token[1]) == 6
    assert set(pred.get_field("labels").tolist()) <= {1}
    # (b) a caption string with one entity per phrase (`run_ner` = nltk noun phrases in the reference; here: the spans directly)
    caption = "".join(n + " . " for n in names)
    spans, pos = [], 0
    for n in names:
        spans.append([[pos, pos + len(n)]])
        pos += len(n) + 3
    demo.run_ner = lambda text: spans
    pred = demo.compute_prediction(image, caption)
    assert sorted(demo.positive_map_label_to_token) == [1, 2, 3, 4]
    assert pred.size == (190, 150) and not pred.has_field("mask")
    top = demo._post_process_fixed_thresh(pred)
    top2 = demo._post_process(pred, threshold=0.06)
    assert len(top) == len(top2) > 0 and bool((top.get_field("scores")[:-1] >= top.get_field("scores")[1:]).all())
    assert float(top.get_field("scores").min()) > 0.06 and set(top.get_field("labels").tolist()) <= {1, 2, 3, 4}

    # oracle on the same pixels / caption / label map
    t = demo.tokenizer([caption], max_length=256, padding="max_length", return_special_tokens_mask=True, return_tensors="pt", truncation=True)
    pimg, sizes = od.pad_images([demo.transforms(image)], cfg.DATALOADER.SIZE_DIVISIBILITY)
    with torch.no_grad():
        dets = od.forward(sd, spec, pimg, sizes, t["input_ids"], t["attention_mask"], dict(demo.positive_map_label_to_token), None)
    keep = dets[0]["scores"] > 0.06
    o1 = torch.argsort(dets[0]["scores"][keep], descending=True, stable=True)
    assert int(keep.sum()) == len(top)
    assert torch.allclose(dets[0]["scores"][keep][o1], top.get_field("scores"), atol=2e-4)
    assert torch.equal(dets[0]["labels"][keep][o1], top.get_field("labels"))
    assert torch.allclose(dets[0]["boxes"][keep][o1], top.bbox, atol=2e-2)
