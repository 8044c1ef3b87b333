"""Oracle: ROIAlign, FPN level mapping, the (Custom)Pooler and vision-query extraction (test infrastructure, see
oracle/__init__.py).

Restates
  * maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:16-123 (bilinear_interpolate + RoIAlignForward, the legacy operator) --
    PINNED on the GPU against that very kernel compiled by oracle/build_ref.py (tests/parity_checks.check_roi_align);
  * `torchvision.ops.roi_align(..., aligned=True)` (third-party, torchvision 0.15.2 pinned by the reference's
    requirements.txt:2; called by layers/roi_align.py:79-81 ROIAlignV2): the published algorithm is the same kernel with the
    box shifted by -0.5 after scaling and WITHOUT the "force malformed ROIs to be 1x1" clamp of the legacy version
    (torchvision/csrc/ops/cuda/roi_align_kernel.cu: `offset = aligned ? 0.5 : 0`, `if (!aligned) roi_w = max(roi_w, 1)`);
  * maskrcnn_benchmark/modeling/poolers.py:11-42 (LevelMapper), :45-128 (Pooler), :130-168 (CustomPooler);
  * maskrcnn_benchmark/modeling/detector/generalized_vl_rcnn_new.py:32-49 (expand_bbox), :232-288 (extract_query).
"""
import math

import torch
import torch.nn.functional as F


def _bilinear(feat, y, x):
    """bilinear_interpolate (ROIAlign_cuda.cu:16-63).  feat [C, H, W]; y, x [...] -> [C, ...]."""
    C, H, W = feat.shape
    outside = (y < -1.0) | (y > H) | (x < -1.0) | (x > W)
    y = y.clamp(min=0)
    x = x.clamp(min=0)
    y_low, x_low = y.floor().long(), x.floor().long()
    top, right = y_low >= H - 1, x_low >= W - 1
    y_low = torch.where(top, torch.full_like(y_low, H - 1), y_low)
    x_low = torch.where(right, torch.full_like(x_low, W - 1), x_low)
    y_high = torch.where(top, y_low, y_low + 1)
    x_high = torch.where(right, x_low, x_low + 1)
    y = torch.where(top, y_low.to(y.dtype), y)
    x = torch.where(right, x_low.to(x.dtype), x)
    ly, lx = y - y_low, x - x_low
    hy, hx = 1.0 - ly, 1.0 - lx
    flat = feat.reshape(C, H * W)

    def at(yy, xx):
        return flat[:, (yy * W + xx).reshape(-1)].reshape(C, *yy.shape)
    val = hy * hx * at(y_low, x_low) + hy * lx * at(y_low, x_high) + ly * hx * at(y_high, x_low) + ly * lx * at(y_high, x_high)
    return torch.where(outside, torch.zeros_like(val), val)


def roi_align(feat, rois, output_size, spatial_scale, sampling_ratio, aligned=False):
    """feat [N, C, H, W] fp32, rois [R, 5] (batch index, x1, y1, x2, y2) -> [R, C, PH, PW]."""
    PH, PW = (output_size, output_size) if isinstance(output_size, int) else output_size
    R = rois.shape[0]
    C = feat.shape[1]
    out = feat.new_zeros(R, C, PH, PW)
    off = 0.5 if aligned else 0.0
    for r in range(R):
        b = int(rois[r, 0])
        x1, y1, x2, y2 = (rois[r, 1:] * spatial_scale - off).tolist()
        rw, rh = x2 - x1, y2 - y1
        if not aligned:                                       # "Force malformed ROIs to be 1x1" (:88-89)
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / PH, rw / PW
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / PH))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / PW))
        count = max(gh * gw, 1) if aligned else gh * gw
        if gh <= 0 or gw <= 0:
            continue                                          # no sample points: the bin sums stay 0 (0 / count)
        ph = torch.arange(PH, dtype=feat.dtype)[:, None, None, None]
        pw = torch.arange(PW, dtype=feat.dtype)[None, :, None, None]
        iy = torch.arange(gh, dtype=feat.dtype)[None, None, :, None]
        ix = torch.arange(gw, dtype=feat.dtype)[None, None, None, :]
        y = (y1 + ph * bh + (iy + 0.5) * bh / gh).expand(PH, PW, gh, gw)
        x = (x1 + pw * bw + (ix + 0.5) * bw / gw).expand(PH, PW, gh, gw)
        out[r] = _bilinear(feat[b], y, x).sum((-1, -2)) / count
    return out


def level_mapper(areas, k_min, k_max, s0=224, lvl0=4, eps=1e-6):
    """poolers.py:11-42 -> level index (0-based) per box."""
    lv = torch.floor(lvl0 + torch.log2(torch.sqrt(areas) / s0 + eps))
    return torch.clamp(lv, min=k_min, max=k_max).to(torch.int64) - int(k_min)


def _rois(boxes_per_image):
    return torch.cat([torch.cat([torch.full((len(b), 1), float(i)), b], 1) for i, b in enumerate(boxes_per_image)])


def pooler(feats, boxes_per_image, output_size, scales, sampling_ratio, every_level=False):
    """Pooler.forward (every_level=False: each box from its mapped level -> [R, C, P, P]) / CustomPooler.forward
    (every_level=True -> [L, R, C, P, P]); ROIAlignV2 = aligned.  boxes: list of [n, 4] xyxy (BoxList.area uses +1)."""
    rois = _rois(boxes_per_image)
    if len(scales) == 1:
        return roi_align(feats[0], rois, output_size, scales[0], sampling_ratio, aligned=True)
    if every_level:
        return torch.stack([roi_align(f, rois, output_size, s, sampling_ratio, aligned=True) for f, s in zip(feats, scales)])
    k_min = -math.log2(scales[0])
    k_max = -math.log2(scales[-1])
    b = rois[:, 1:]
    areas = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)                     # BoxList.area, xyxy: TO_REMOVE = 1
    levels = level_mapper(areas, k_min, k_max)
    out = feats[0].new_zeros(len(rois), feats[0].shape[1], output_size, output_size)
    for lvl, (f, s) in enumerate(zip(feats, scales)):
        idx = torch.nonzero(levels == lvl).squeeze(1)
        if len(idx):
            out[idx] = roi_align(f, rois[idx], output_size, s, sampling_ratio, aligned=True)
    return out


def expand_bbox(boxes, image_size, labels, ratio=1.5):
    """generalized_vl_rcnn_new.py:32-49 (+ BoxList.clip_to_image(remove_empty=True), bounding_box.py:221-232).
    boxes [n,4] xyxy, image_size (w, h) -> (boxes, labels) of the kept boxes."""
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    dw, dh = (w * ratio - w) / 2, (h * ratio - h) / 2
    nb = boxes + torch.stack([-dw, -dh, dw, dh], 1)
    W, H = image_size
    nb = torch.stack([nb[:, 0].clamp(0, W - 1), nb[:, 1].clamp(0, H - 1), nb[:, 2].clamp(0, W - 1), nb[:, 3].clamp(0, H - 1)], 1)
    keep = (nb[:, 3] > nb[:, 1]) & (nb[:, 2] > nb[:, 0])
    return nb[keep], labels[keep]


def extract_query(feats, targets, query_images, spec_pool, select_fpn_level=True, expand_ratio=1.5, exclude_similar=False,
                  max_query_number=5000, similarity_threshold=0.85):
    """generalized_vl_rcnn_new.py:232-288.  feats: FPN maps [B,C,H,W]; targets: list of (boxes xyxy [n,4], labels [n],
    (w, h)); query_images: dict label -> [n, scales, C] (or [] / missing); spec_pool = (resolution, scales, sampling).
    Returns the updated bank (same object)."""
    res, scales, sampling = spec_pool
    exp = [expand_bbox(b, size, l, expand_ratio) for (b, l, size) in targets]
    q = pooler(feats, [e[0] for e in exp], res, scales, sampling, every_level=not select_fpn_level)
    if select_fpn_level:
        q = q[None]
    q = q.mean(dim=[-2, -1]).permute(1, 0, 2)                                       # [boxes, scales, C]
    labels = torch.cat([e[1] for e in exp])
    for label, feat in zip(labels.tolist(), q):
        cur = query_images.get(label, [])
        n = len(cur)
        if n >= max_query_number:
            continue
        if exclude_similar and n > 0:
            bank = F.normalize(cur, p=2, dim=-1)
            new = F.normalize(feat, p=2, dim=-1)
            sim = torch.einsum("bnd,nd->bn", bank, new)
            if (sim > similarity_threshold).sum() > 0:
                continue
        query_images[label] = feat[None] if n == 0 else torch.cat([cur, feat[None]])
    return query_images
