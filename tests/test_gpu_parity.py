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
    bad = [r for r in res if not r["ok"]]
    assert not bad, "\n".join(f"{r['name']}: max_err={r['max_err']:.3e} norm={r['norm_err']:.3e} tol={r['tol']}" for r in bad)


ATTN = [
    dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True),
    dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True, clamp=50000.0, big=True),
    dict(B=1, H=8, D=32, Nq=200, Nk=5577, nsplit=4),
    dict(B=1, H=8, D=32, Nq=37, Nk=61),
    dict(B=2, H=8, D=32, Nq=1, Nk=9),
    dict(B=1, H=8, D=256, Nq=1500, Nk=256, mask=True, clamp=50000.0, scale=1.0 / 16),
    dict(B=1, H=8, D=256, Nq=256, Nk=1500, clamp=50000.0, scale=1.0 / 16, nsplit=3),
    dict(B=1, H=2, D=256, Nq=130, Nk=22400, scale=1.0 / 16, nsplit=8),
    dict(B=1, H=8, D=256, Nq=22400, Nk=256, mask=True, scale=1.0 / 16),
]


@pytest.mark.parametrize("cfg", ATTN, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_attention(dev, cfg):
    import parity_checks as pc
    _assert(pc.check_attention(dev, **cfg))


def test_attention_strided_views(dev):
    import parity_checks as pc
    _assert(pc.check_attention_strided(dev))


@pytest.mark.parametrize("name", ["check_window_attention", "check_swin_fpn", "check_gcp_block", "check_pre_select",
                                  "check_vl_fuse", "check_dcn", "check_dyconv", "check_nms", "check_full_model"])
def test_block(dev, name):
    import parity_checks as pc
    _assert(getattr(pc, name)(dev))


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
