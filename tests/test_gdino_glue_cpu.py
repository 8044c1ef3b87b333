"""MQ-GroundingDINO product pipeline (mq_det_amd/modeling/gdino_pipeline.py) with the HIP entry points replaced by the pure-torch
emulations of tests/ops_emulation.py, in fp32, against the oracle (oracle/gdino.py, pinned to the reference's own module by
tests/test_oracle_golden.py).  Validates on the CPU-only build box every piece of host-side glue: weight folding of the fusion
layers, batched decoder projections, mask / geometry construction, sub-sentence text masks, strides handed to the kernels.
The kernels themselves are covered by the -m gpu tests."""
import os
import sys
import types

import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import ops_emulation as emu  # noqa: E402

from mq_det_amd.config import get_gdino_cfg  # noqa: E402
from mq_det_amd.modeling import gdino_pipeline as gp, pipeline  # noqa: E402
from mq_det_amd.modeling.params import gdino_swin_cfg  # noqa: E402
from mq_det_amd.modeling.query_selector import QuerySelector  # noqa: E402
from oracle import gdino as og  # noqa: E402
from oracle.spec import tiny_gdino_spec  # noqa: E402
from oracle.weights import make_gdino_state_dict, make_query_bank  # noqa: E402


def gdino_cfg(spec, tok_dir=None):
    cfg = get_gdino_cfg()
    G = cfg.GROUNDINGDINO
    G.swin_depths, G.enc_layers, G.dec_layers, G.num_queries = spec.swin_depths, spec.enc_layers, spec.dec_layers, spec.num_queries
    cfg.MODEL.LANGUAGE_BACKBONE.NUM_HIDDEN_LAYERS, cfg.MODEL.LANGUAGE_BACKBONE.QV_START = spec.bert_layers, spec.qv_start
    cfg.MODEL.LANGUAGE_BACKBONE.BERT_VOCAB_SIZE = spec.vocab
    cfg.MODEL.DYHEAD.NUM_CLASSES = spec.num_classes
    cfg.VISION_QUERY.ENABLED = spec.vision_query
    cfg.VISION_QUERY.NUM_QUERY_PER_CLASS = spec.num_query_per_class
    if tok_dir:
        G.text_encoder_type = tok_dir
    return cfg


@pytest.fixture()
def emulated_ops(monkeypatch):
    from mq_det_amd import ops as real
    fake = emu.namespace(real)
    monkeypatch.setattr(pipeline, "ops", fake)
    monkeypatch.setattr(gp, "ops", fake)
    return fake


def close(a, b, tol=5e-4):
    a, b = a.float(), b.float()
    err = (a - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), f"max err {err} vs scale {b.abs().max().item()}"


def make_case(spec, B, sizes, seed=0):
    g = torch.Generator().manual_seed(seed)
    H, W = 128, 160
    img = torch.zeros(B, 3, H, W)
    for i, (h, w) in enumerate(sizes):
        img[i, :, :h, :w] = torch.randn(3, h, w, generator=g)
    # caption "a b . c . d e f ." as ids: [CLS] words... '.' ... [SEP] pad; specials: 101, 102, '.'=1012?, '?'
    special = [101, 102, 1012, 1029]
    ids = torch.zeros(B, 512, dtype=torch.long)
    words, pmap, pos = (2, 1, 3, 1, 2), {}, 1
    row = [101]
    for c, n in enumerate(words):
        pmap[c + 1] = list(range(pos, pos + n))
        row += [1100 + 7 * c + j for j in range(n)] + [1012]
        pos += n + 1
    row += [102]
    ids[:, :len(row)] = torch.tensor(row)
    am = (ids != 0).long()
    return img, ids, am, pmap, special


@pytest.mark.parametrize("vq,B,sizes", [(True, 1, [(120, 150)]), (False, 2, [(128, 130), (100, 160)]), (False, 1, [(128, 160)])])
def test_gdino_pipeline_glue(emulated_ops, vq, B, sizes):
    spec = tiny_gdino_spec(vision_query=vq)
    sd = make_gdino_state_dict(tiny_gdino_spec(), seed=0)
    cfg = gdino_cfg(spec)
    SW = gdino_swin_cfg(cfg)
    dev = torch.device("cpu")
    P = gp.build_gdino_plan(sd, cfg, dev, SW, dtype=torch.float32)
    img, ids, am, pmap, special = make_case(spec, B, sizes)
    bank = make_query_bank(sorted(pmap), spec, seed=1, scales=1) if vq else None
    with torch.no_grad():
        o = og.forward(sd, spec, img, sizes, ids, am, pmap, special, bank)
        geo = gp.geometry(P, cfg, img.shape[2], img.shape[3], sizes, dev)
        txt, max_kv = gp.text_inputs(cfg, ids, am, special, dev)
        vision = idx = None
        if vq:
            qs = QuerySelector(cfg)
            qs.query_bank = bank
            labels = [k for k, v in pmap.items() if len(v)]
            vision, idx = qs.select([labels] * B, [pmap] * B, 256, dev, torch.float32)
        T, C = 256, spec.num_classes - 1
        cmap = torch.zeros(T, C)
        for lab, toks in pmap.items():
            cmap[toks, lab - 1] = 1.0 / len(toks)
        im_hw = torch.tensor(sizes, dtype=torch.float32)
        trace = {}
        out = gp.forward_device(P, cfg, SW, img.contiguous(memory_format=torch.channels_last), geo, txt, vision, idx, cmap, im_hw,
                                max_kv=max_kv, trace=trace)
    # geometry
    assert torch.equal(geo["mask"], o["mask"])
    close(geo["pos"], torch.cat([p_.flatten(2).transpose(1, 2) + sd["transformer.level_embed"][l] for l, p_ in enumerate(o["pos"])], 1))
    # stages
    close(trace["srcs"], torch.cat([s.flatten(2).transpose(1, 2) for s in o["srcs"]], 1))
    n_real = int(am[0].sum())
    close(trace["bert"], o["bert"])
    close(trace["encoded_text"], o["encoded_text"])
    close(trace["memory"], o["memory"])
    # rows of padding tokens are not comparable (the text-side kernel skips 16-row blocks of pure padding); nothing reads them:
    # masked as keys everywhere, -inf in the contrastive logits
    close(trace["memory_text"][:, :n_real], o["memory_text"][:, :n_real])
    assert torch.equal(trace["topk"], o["topk"])
    close(trace["init_box"], o["init_box"])
    close(trace["hs_enc"], o["hs_enc"])
    for a, b in zip(trace["refs"], o["refs"]):
        close(a, b)
    close(trace["hs"], o["hs"][-1])
    close(trace["pred_logits"], o["pred_logits"])
    close(trace["pred_boxes"], o["pred_boxes"])
    for b in range(B):
        bx, sc, lb = o["detections"][b]
        keep = out["keep"][b]
        assert int(keep.sum()) == len(bx) > 0
        sel = out["packed"][b][keep]
        close(sel[:, :4], bx, tol=1e-4)
        close(sel[:, 4], sc)
        assert torch.equal(sel[:, 5].long(), lb)
    assert n_real < 256


def test_text_enhancer_mask_quirk_for_different_captions():
    """B > 1 with different captions: head (b, h) of the text enhancer reads the mask of batch element (b*nhead + h) % B."""
    spec = tiny_gdino_spec()
    cfg = gdino_cfg(spec)
    ids = torch.zeros(2, 512, dtype=torch.long)
    ids[0, :8] = torch.tensor([101, 5, 6, 1012, 7, 1012, 102, 0])
    ids[1, :8] = torch.tensor([101, 5, 1012, 6, 7, 8, 1012, 102])
    txt, max_kv = gp.text_inputs(cfg, ids, (ids != 0).long(), [101, 102, 1012, 1029], torch.device("cpu"))
    attn, _ = og.special_token_masks(ids, [101, 102, 1012, 1029])
    rep = (~attn[:, :256, :256]).repeat(4, 1, 1).view(2, 4, 256, 256)          # what torch's MHA sees: index b*4 + h
    assert torch.equal(txt["enh_mask"].bool(), rep)
    assert max_kv == 8


def test_convert_marks_dropped_queries_for_the_cross_rank_gather():
    """`convert` keeps the [B, nq, 6] shape (one fixed-shape all-gather across ranks): dropped queries get score -1, which is what
    parallel.unpack_detections filters on; kept rows equal the oracle's conversion."""
    from mq_det_amd import parallel
    g = torch.Generator().manual_seed(3)
    prob = torch.sigmoid(torch.randn(2, 30, 256, generator=g) * 2.0 - 5.0)
    boxes = torch.rand(2, 30, 4, generator=g) * torch.tensor([1.0, 1.0, 0.6, 0.6])
    pmap = {1: [1, 2], 2: [4], 5: [6, 7, 8]}
    cmap = torch.zeros(256, 80)
    for lab, toks in pmap.items():
        cmap[toks, lab - 1] = 1.0 / len(toks)
    sizes = [(128, 130), (100, 160)]
    packed, keep = gp.convert(prob, boxes, cmap, False, torch.tensor(sizes, dtype=torch.float32), 0.05)
    ref = og.convert_to_glip_output(prob, boxes, pmap, sizes, 81, 0.05)
    dets = parallel.unpack_detections(packed)
    for b in range(2):
        assert 0 < int(keep[b].sum()) < 30 and bool((packed[b, ~keep[b], 4] == -1).all())
        close(dets[b]["boxes"], ref[b][0], tol=1e-5)
        close(dets[b]["scores"], ref[b][1], tol=1e-6)
        assert torch.equal(dets[b]["labels"], ref[b][2])
    packed, keep = gp.convert(prob, boxes, cmap, True, torch.tensor(sizes, dtype=torch.float32), 0.05)    # empty-label quirk
    assert not bool(keep.any()) and all(len(d["boxes"]) == 0 for d in parallel.unpack_detections(packed))


def test_gdino_pipeline_two_different_captions(emulated_ops):
    """B = 2 with different captions (text only): the product's expanded [B, heads, T, T] text-enhancer mask reproduces the
    reference's `mask.repeat(nhead)` indexing, end to end against the oracle (pinned to the reference module on this very
    situation by tests/test_oracle_golden.py::...[gdino_text2])."""
    spec = tiny_gdino_spec(vision_query=False)
    sd = make_gdino_state_dict(tiny_gdino_spec(), seed=0)
    cfg = gdino_cfg(spec)
    SW = gdino_swin_cfg(cfg)
    dev = torch.device("cpu")
    P = gp.build_gdino_plan(sd, cfg, dev, SW, dtype=torch.float32)
    sizes = [(128, 130), (100, 160)]
    img, ids, am, pmap, special = make_case(spec, 2, sizes)
    row = [101, 1300, 1012, 1301, 1302, 1303, 1012, 1304, 1305, 1012, 1306, 1012, 102]      # other sub-sentence lengths
    ids[1] = 0
    ids[1, :len(row)] = torch.tensor(row)
    am = (ids != 0).long()
    with torch.no_grad():
        o = og.forward(sd, spec, img, sizes, ids, am, pmap, special, None)
        geo = gp.geometry(P, cfg, img.shape[2], img.shape[3], sizes, dev)
        txt, max_kv = gp.text_inputs(cfg, ids, am, special, dev)
        assert txt["enh_mask"].shape == (2, 4, 256, 256) and not torch.equal(txt["enh_mask"][0, 0], txt["enh_mask"][0, 1])
        cmap = torch.zeros(256, spec.num_classes - 1)
        for lab, toks in pmap.items():
            cmap[toks, lab - 1] = 1.0 / len(toks)
        trace = {}
        gp.forward_device(P, cfg, SW, img.contiguous(memory_format=torch.channels_last), geo, txt, None, None, cmap,
                          torch.tensor(sizes, dtype=torch.float32), max_kv=max_kv, trace=trace)
    for b in range(2):
        n = int(am[b].sum())
        close(trace["memory_text"][b, :n], o["memory_text"][b, :n])
    close(trace["memory"], o["memory"])
    assert torch.equal(trace["topk"], o["topk"])
    close(trace["pred_logits"], o["pred_logits"])
    close(trace["pred_boxes"], o["pred_boxes"])
