"""Oracle: anchors, ATSS post-processing, class-aware NMS (test infrastructure, see oracle/__init__.py).

Restates
  * maskrcnn_benchmark/modeling/rpn/anchor_generator.py:73-95,157-181,356-425 (grid anchors),
  * maskrcnn_benchmark/modeling/rpn/vldyhead.py:78-108 (BoxCoder.decode),
  * maskrcnn_benchmark/modeling/rpn/inference.py:620-712 (per-level), :714-769 (merge, NMS, top-k),
    :772-824 (token -> class score aggregation, MEAN),
  * maskrcnn_benchmark/structures/bounding_box.py:221-232 (clip_to_image),
  * maskrcnn_benchmark/csrc/cuda/ml_nms.cu:15-26 (IoU with legacy +1, 0 across labels),
    :78-149 (sort by score, greedy sweep, kept indices returned in ascending original order).
"""
import math

import numpy as np
import torch


def cell_anchor(stride, size):
    """generate_anchors(stride, (size,), (1.0,)) -- anchor_generator.py:356-425, one square anchor."""
    w = h = float(stride)
    ctr = 0.5 * (w - 1)
    ws = np.round(np.sqrt(w * h / 1.0))
    hs = np.round(ws * 1.0)
    scale = size / stride
    ws, hs = ws * scale, hs * scale
    return torch.tensor([[ctr - 0.5 * (ws - 1), ctr - 0.5 * (hs - 1), ctr + 0.5 * (ws - 1), ctr + 0.5 * (hs - 1)]],
                        dtype=torch.float32)


def grid_anchors(grid_sizes, spec):
    """anchor_generator.py:73-95: per level [H*W, 4] (1 anchor per location, y-major)."""
    out = []
    for (H, W), stride, size in zip(grid_sizes, spec.anchor_strides, spec.anchor_sizes):
        sx = torch.arange(0, W * stride, step=stride, dtype=torch.float32)
        sy = torch.arange(0, H * stride, step=stride, dtype=torch.float32)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        shifts = torch.stack([xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)], 1)
        out.append((shifts[:, None, :] + cell_anchor(stride, size)[None]).reshape(-1, 4))
    return out


def box_decode(preds, anchors):
    """vldyhead.py:78-108 (weights 10,10,5,5; legacy +1 widths; dw/dh clamped at log(1000/16))."""
    w = anchors[:, 2] - anchors[:, 0] + 1
    h = anchors[:, 3] - anchors[:, 1] + 1
    cx = (anchors[:, 2] + anchors[:, 0]) / 2
    cy = (anchors[:, 3] + anchors[:, 1]) / 2
    dx, dy = preds[:, 0] / 10.0, preds[:, 1] / 10.0
    lim = math.log(1000.0 / 16)
    dw, dh = (preds[:, 2] / 5.0).clamp(max=lim), (preds[:, 3] / 5.0).clamp(max=lim)
    pcx, pcy = dx * w + cx, dy * h + cy
    pw, ph = torch.exp(dw) * w, torch.exp(dh) * h
    return torch.stack([pcx - 0.5 * (pw - 1), pcy - 0.5 * (ph - 1), pcx + 0.5 * (pw - 1), pcy + 0.5 * (ph - 1)], 1)


def token_to_class_scores(prob, positive_map, num_class, minus_one=True, score_agg="MEAN"):
    """convert_grounding_to_od_logits(_v2) (inference.py:772-824), every aggregation the reference has: MEAN, MAX, POWER
    (geometric mean; the MDETR-style _v2 only), ONEHOT (class column j = token j: the first len(positive_map) token columns).
    prob: [B, HW, T] sigmoid-ed; positive_map: {label(1-based): [token idx]} -> [B, HW, num_class] ([B, HW, len(map)] for ONEHOT)."""
    if score_agg == "ONEHOT":
        return prob[:, :, :len(positive_map)]
    if score_agg not in ("MEAN", "MAX", "POWER"):
        raise NotImplementedError(score_agg)
    scores = torch.zeros(prob.shape[0], prob.shape[1], num_class)
    for label, toks in positive_map.items():
        if isinstance(toks, int):
            toks = [toks]
        sel = prob[:, :, torch.tensor(toks, dtype=torch.long)]
        if score_agg == "MEAN":
            v = sel.mean(-1)
        elif score_agg == "MAX":
            v = sel.max(-1)[0]
        else:
            v = torch.pow(torch.prod(sel, dim=-1), 1 / len(toks))
        scores[:, :, label - 1 if minus_one else label] = v
    return scores


def ml_nms(boxes, scores, labels, thresh):
    """ml_nms.cu: greedy class-aware NMS.  Returns kept indices (ascending original order)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long)
    order = torch.sort(scores, descending=True, stable=True)[1]
    b, lab = boxes[order], labels[order]
    area = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    lt = torch.max(b[:, None, :2], b[None, :, :2])
    rb = torch.min(b[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    iou = inter / (area[:, None] + area[None, :] - inter)
    iou = torch.where(lab[:, None] == lab[None, :], iou, torch.zeros_like(iou))
    over = (iou > thresh).numpy()
    removed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if not removed[i]:
            keep.append(i)
            removed[i + 1:] |= over[i, i + 1:]
    return torch.sort(order[torch.tensor(keep, dtype=torch.long)])[0]


def atss_postprocess(bbox_reg, centerness, logits, anchors, image_sizes, positive_map, spec):
    """ATSSPostProcessor.forward (inference.py:714-769).  Returns per image a dict
    {boxes [n,4] xyxy, scores [n], labels [n] (1-based)} plus the pre-NMS candidates."""
    B = bbox_reg[0].shape[0]
    per_image = [[] for _ in range(B)]
    for reg, ctr, logit, anc in zip(bbox_reg, centerness, logits, anchors):
        _, _, H, W = reg.shape
        prob = logit.sigmoid()
        if spec.mdetr_class_num != -1:
            cls = token_to_class_scores(prob, positive_map, spec.mdetr_class_num, minus_one=True, score_agg=spec.score_agg)
        else:
            if spec.score_agg == "POWER":
                raise NotImplementedError("POWER exists in convert_grounding_to_od_logits_v2 only (inference.py:775-792)")
            cls = token_to_class_scores(prob, positive_map, spec.num_classes - 1, minus_one=True, score_agg=spec.score_agg)
        reg = reg.permute(0, 2, 3, 1).reshape(B, -1, 4)
        cand = cls > spec.pre_nms_thresh
        topn = cand.reshape(B, -1).sum(1).clamp(max=spec.pre_nms_top_n)
        cls = cls * ctr.permute(0, 2, 3, 1).reshape(B, -1).sigmoid()[:, :, None]
        for b in range(B):
            sc = cls[b][cand[b]]
            sc, top = sc.topk(int(topn[b]), sorted=False)
            nz = cand[b].nonzero()[top]
            loc, lab = nz[:, 0], nz[:, 1] + 1
            det = box_decode(reg[b][loc], anc[loc])
            h, w = image_sizes[b]
            det = torch.stack([det[:, 0].clamp(0, w - 1), det[:, 1].clamp(0, h - 1),
                               det[:, 2].clamp(0, w - 1), det[:, 3].clamp(0, h - 1)], -1)
            wh_ok = ((det[:, 2] - det[:, 0] + 1) >= 0) & ((det[:, 3] - det[:, 1] + 1) >= 0)  # min_size 0
            per_image[b].append((det[wh_ok], torch.sqrt(sc)[wh_ok], lab[wh_ok]))
    results = []
    for b in range(B):
        boxes = torch.cat([x[0] for x in per_image[b]])
        scores = torch.cat([x[1] for x in per_image[b]])
        labels = torch.cat([x[2] for x in per_image[b]])
        pre = {"boxes": boxes, "scores": scores, "labels": labels}
        keep = ml_nms(boxes, scores, labels.float(), spec.nms_thresh)
        boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
        n = len(keep)
        if n > spec.detections_per_img > 0:
            thr = torch.kthvalue(scores, n - spec.detections_per_img + 1)[0]
            k = (scores >= thr).nonzero().squeeze(1)
            boxes, scores, labels = boxes[k], scores[k], labels[k]
        results.append({"boxes": boxes, "scores": scores, "labels": labels, "pre_nms": pre})
    return results
