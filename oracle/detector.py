"""Oracle: GeneralizedVLRCNN_New eval forward (test infrastructure, see oracle/__init__.py).

Restates maskrcnn_benchmark/modeling/detector/generalized_vl_rcnn_new.py:291-305 (pooled FPN
tokens, label -> token maps), :332-455 (eval branch of forward) and
maskrcnn_benchmark/modeling/query_selector/query_selector.py:40-116 (vision-query selection, eval
mode: `sorted(np.random.choice(len, min(len, k), replace=False))` rows of each label, `pad_sequence`, 0/1 mask).
The reference asserts B == 1 when vision queries are on (:354); like the product we lift it by
repeating the (identical) caption maps for every image of the batch.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .backbone import swin_forward, fpn_forward
from .language import language_backbone
from .head import vldyhead
from .postprocess import grid_anchors, atss_postprocess


def pad_images(images, divisible):
    """structures/image_list.py:29-70: zero-pad a list of [3,h,w] to a common /divisible size."""
    hs, ws = [im.shape[1] for im in images], [im.shape[2] for im in images]
    H, W = max(hs), max(ws)
    if divisible > 0:
        H = -(-H // divisible) * divisible
        W = -(-W // divisible) * divisible
    out = torch.zeros(len(images), 3, H, W)
    for i, im in enumerate(images):
        out[i, :, :im.shape[1], :im.shape[2]] = im
    return out, [(int(h), int(w)) for h, w in zip(hs, ws)]


def labels_and_maps(positive_map, T):
    """generalized_vl_rcnn_new.py:295-305."""
    labels = [k for k, v in positive_map.items() if len(v) != 0]
    m = torch.zeros(len(labels), T)
    for j, lab in enumerate(labels):
        m[j, positive_map[lab]] = 1
    return labels, m / (m.sum(-1)[:, None] + 1e-6)


def select_queries(bank, batched_labels, batched_maps, k):
    """query_selector.py:40-116 (eval).  bank: {label: [n, scales, C]} -> vision [B,V,C], mask [B,V,T]."""
    if bank is None:
        return None, None
    qs, ms = [], []
    for labels, maps in zip(batched_labels, batched_maps):
        q_img, m_img = [], []
        for lab, loc in zip(labels, maps):
            cand = bank[lab]
            n = min(len(cand), k)
            # reference (:74-78): idx = sorted(np.random.choice(len, n, replace=False)); a `[]` entry of the
            # defaultdict(list) bank contributes nothing.  The draw is the identity when the bank holds <= k rows.
            idx = sorted(np.random.choice(len(cand), n, replace=False).tolist())
            if isinstance(cand, list):
                assert len(idx) == 0
                continue
            q = cand[idx]
            scales = q.shape[1]
            q_img.append(q.flatten(0, 1))
            m_img.append(loc[None].expand(n * scales, -1))
        qs.append(torch.cat(q_img))
        ms.append(torch.cat(m_img))
    vision = torch.nn.utils.rnn.pad_sequence(qs, batch_first=True)
    mask = torch.nn.utils.rnn.pad_sequence(ms, batch_first=True).clone()
    mask[mask != 0] = 1
    return vision, mask


def pooled_fpn_tokens(feats):
    """generalized_vl_rcnn_new.py:291-293: AvgPool2d(2) per level, flatten, concat -> [B, sum, C]."""
    return torch.cat([F.avg_pool2d(f, 2).flatten(2) for f in feats], 2).permute(0, 2, 1)


@torch.no_grad()
def forward(sd, spec, images, image_sizes, input_ids, attention_mask, positive_map, bank=None,
            return_intermediates=False):
    """images: [B,3,H,W] already padded; input_ids/attention_mask: [B,T] int64.
    Returns list of per-image dicts (boxes/scores/labels) -- and every intermediate when asked."""
    B = images.shape[0]
    c = swin_forward(sd, "backbone.body", images, spec)
    feats = fpn_forward(sd, "backbone.fpn", c)
    vision = vmask = pooled = None
    if spec.vision_query and bank is not None:
        labels, amap = labels_and_maps(positive_map, spec.max_query_len)
        vision, vmask = select_queries(bank, [labels] * B, [amap] * B, spec.num_query_per_class)
        pooled = pooled_fpn_tokens(feats)
    lang = language_backbone(sd, "language_backbone.body", input_ids, attention_mask, vision, pooled, vmask, spec)
    trace = [] if return_intermediates else None
    head = vldyhead(sd, "rpn.head", feats, dict(lang), spec, trace=trace)
    anchors = grid_anchors([f.shape[-2:] for f in feats], spec)
    dets = atss_postprocess(head["bbox_reg"], head["centerness"], head["dot_product_logits"], anchors,
                            image_sizes, positive_map, spec)
    if return_intermediates:
        return dets, {"swin": c, "fpn": feats, "lang": lang, "head": head, "head_trace": trace, "anchors": anchors,
                      "vision": vision, "vision_mask": vmask, "pooled": pooled}
    return dets
