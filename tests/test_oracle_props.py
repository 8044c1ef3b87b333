"""Known-answer / property checks for the two CUDA-only reference ops the oracle restates from .cu
sources (DCNv2 forward, ml_nms) -- the reference has no CPU implementation or fixtures for them."""
import torch
import torch.nn.functional as F

from oracle.head import dcn_v2, bilinear_zero
from oracle.postprocess import ml_nms, box_decode


def test_dcn_zero_offset_is_conv():
    torch.manual_seed(0)
    x = torch.randn(2, 8, 9, 11)
    w, b = torch.randn(6, 8, 3, 3), torch.randn(6)
    for stride in (1, 2):
        Ho, Wo = (9 + 2 - 3) // stride + 1, (11 + 2 - 3) // stride + 1
        off, msk = torch.zeros(2, 18, Ho, Wo), torch.ones(2, 9, Ho, Wo)
        torch.testing.assert_close(dcn_v2(x, off, msk, w, b, stride), F.conv2d(x, w, b, stride=stride, padding=1),
                                   atol=1e-5, rtol=1e-5)


def test_dcn_integer_offset_is_shift_and_mask_scales():
    torch.manual_seed(1)
    x = torch.randn(1, 4, 8, 8)
    w = torch.randn(3, 4, 3, 3)
    off = torch.zeros(1, 18, 8, 8)
    off[:, 0::2] = 1.0      # dh = +1 for every tap
    off[:, 1::2] = -2.0     # dw = -2
    msk = torch.full((1, 9, 8, 8), 0.5)
    xs = torch.zeros_like(x)
    xs[:, :, :-1, 2:] = x[:, :, 1:, :-2]          # xs[h, w] = x[h+1, w-2], zero outside
    # interior outputs only: at the border the conv zero-pads the SHIFTED image while DCN samples the original
    got, want = dcn_v2(x, off, msk, w, None, 1), 0.5 * F.conv2d(xs, w, None, padding=1)
    torch.testing.assert_close(got[:, :, 1:7, 1:7], want[:, :, 1:7, 1:7], atol=1e-5, rtol=1e-5)


def test_dcn_flat_offset_indexing_quirk():
    """Offsets computed on a bigger map (level l) applied to a conv whose output is smaller (level l+1):
    the kernel reads the buffer flat with the OUTPUT dims (deform_conv_kernel_cuda.cu:607-617)."""
    torch.manual_seed(2)
    x = torch.randn(1, 4, 5, 6)                  # level l+1 input, output 5x6
    w = torch.randn(2, 4, 3, 3)
    off_big = torch.randn(1, 18, 10, 12)         # produced from level l (10x12)
    msk_big = torch.rand(1, 9, 10, 12)
    n = 30
    off_small = off_big.reshape(1, -1)[:, :18 * n].reshape(1, 18, 5, 6)
    msk_small = msk_big.reshape(1, -1)[:, :9 * n].reshape(1, 9, 5, 6)
    torch.testing.assert_close(dcn_v2(x, off_big, msk_big, w, None, 1), dcn_v2(x, off_small, msk_small, w, None, 1))


def test_bilinear_zero_border():
    x = torch.arange(12.0).reshape(1, 1, 3, 4)
    h = torch.tensor([[-1.0, -0.5, 2.5, 3.0, 1.0]])
    w = torch.tensor([[0.0, 0.0, 3.0, 0.0, 3.5]])
    v = bilinear_zero(x, h, w)[0, 0]
    assert v[0] == 0 and v[3] == 0
    assert abs(v[1] - 0.0) < 1e-6                # 0.5*x[0,0] = 0
    assert abs(v[2] - 0.5 * 11) < 1e-6           # half of the bottom-right pixel
    assert abs(v[4] - 0.5 * 7) < 1e-6


def _brute_nms(boxes, scores, labels, thr):
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))
    keep = []
    for i in order:
        ok = True
        for j in keep:
            if labels[i] != labels[j]:
                continue
            a, b = boxes[i], boxes[j]
            iw = max(min(a[2], b[2]) - max(a[0], b[0]) + 1, 0)
            ih = max(min(a[3], b[3]) - max(a[1], b[1]) + 1, 0)
            inter = iw * ih
            ua = (a[2] - a[0] + 1) * (a[3] - a[1] + 1) + (b[2] - b[0] + 1) * (b[3] - b[1] + 1) - inter
            if inter / ua > thr:
                ok = False
                break
        if ok:
            keep.append(i)
    return sorted(keep)


def test_ml_nms_matches_bruteforce():
    torch.manual_seed(3)
    n = 200
    xy = torch.rand(n, 2) * 60
    wh = torch.rand(n, 2) * 30 + 2
    boxes = torch.cat([xy, xy + wh], 1)
    scores = torch.rand(n)
    labels = torch.randint(1, 4, (n,)).float()
    keep = ml_nms(boxes, scores, labels, 0.6)
    assert keep.tolist() == _brute_nms(boxes.tolist(), scores.tolist(), labels.tolist(), 0.6)
    assert ml_nms(boxes[:0], scores[:0], labels[:0], 0.6).numel() == 0


def test_box_decode_identity_and_clamp():
    anchors = torch.tensor([[-28.0, -28.0, 35.0, 35.0]])
    out = box_decode(torch.zeros(1, 4), anchors)
    torch.testing.assert_close(out, anchors)
    big = box_decode(torch.tensor([[0.0, 0.0, 1e4, 1e4]]), anchors)
    assert abs(float(big[0, 2] - big[0, 0]) + 1 - 64 * 1000 / 16) < 1e-2


def test_pad_max_false_gives_the_same_detections():
    """LANGUAGE_BACKBONE.PAD_MAX = False (captions padded to the longest of the batch, generalized_vl_rcnn_new.py:378-383) vs
    True (padded to MAX_QUERY_LEN): padding is masked everywhere, so the reference's detections do not depend on it -- the reason
    the product pads to MAX_QUERY_LEN for both settings (detector.tokenize).  Text-only path: with vision queries the reference
    itself needs T == MAX_QUERY_LEN (its query masks are built [.., MAX_QUERY_LEN], generalized_vl_rcnn_new.py:295-305)."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import parity_checks as pc
    from oracle import tiny_spec, detector as od
    from oracle.weights import make_state_dict
    spec = tiny_spec()
    sd = make_state_dict(spec, 0)
    images, sizes, ids, am, pm, bank = pc.make_inputs(spec)
    n = int(am[0].sum())
    with torch.no_grad():
        full = od.forward(sd, spec, images, sizes, ids, am, pm, None)
        short = od.forward(sd, spec, images, sizes, ids[:, :n], am[:, :n], pm, None)
    for a, b in zip(full, short):
        assert len(a["boxes"]) == len(b["boxes"]) > 0
        assert torch.allclose(a["scores"], b["scores"], atol=2e-5) and torch.equal(a["labels"], b["labels"])
        assert torch.allclose(a["boxes"], b["boxes"], atol=1e-2)
