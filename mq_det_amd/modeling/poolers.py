"""Box pooler of the vision-query extraction path (reference modeling/poolers.py:11-168, same class names, constructor
arguments and forward contracts) on the HIP ROIAlign kernel (mq_roi_align_fwd).

`Pooler` maps every box to one FPN level (LevelMapper, FPN paper eq. 1) and pools it there; `CustomPooler` pools every
box on every level (`VISION_QUERY.SELECT_FPN_LEVEL = False`).  Both use the aligned operator (`use_v2=True`, ROIAlignV2 =
torchvision.ops.roi_align(aligned=True)) like generalized_vl_rcnn_new.py:108-121; `use_v2=False` gives the legacy one.
MI355X-first: features may be handed over in the product's native layout (NHWC memory viewed as [B, C, H, W], fp16) -- the
kernel reads arbitrary strides, 64 consecutive channels per wave -- and `forward(..., reduce_mean=True)` returns the bin
average [R, C] directly (the only thing extract_query keeps, generalized_vl_rcnn_new.py:263) instead of [R, C, P, P]."""
import math

import torch
from torch import nn

from .. import ops


class LevelMapper:
    def __init__(self, k_min, k_max, canonical_scale=224, canonical_level=4, eps=1e-6):
        self.k_min, self.k_max, self.s0, self.lvl0, self.eps = k_min, k_max, canonical_scale, canonical_level, eps

    def __call__(self, boxlists):
        s = torch.sqrt(torch.cat([b.area() for b in boxlists]))
        lv = torch.floor(self.lvl0 + torch.log2(s / self.s0 + self.eps))
        return torch.clamp(lv, min=self.k_min, max=self.k_max).to(torch.int64) - int(self.k_min)


class Pooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio, use_v2=False):
        super().__init__()
        self.output_size = output_size if isinstance(output_size, (tuple, list)) else (output_size, output_size)
        self.scales = tuple(float(s) for s in scales)
        self.sampling_ratio = sampling_ratio
        self.aligned = bool(use_v2)
        self.map_levels = LevelMapper(-math.log2(self.scales[0]), -math.log2(self.scales[-1]))

    @staticmethod
    def convert_to_roi_format(boxes):
        """[(K1, 4), (K2, 4)] BoxLists -> [K1 + K2, 5] (batch index first)."""
        return torch.cat([torch.cat([torch.full((len(b), 1), float(i), dtype=torch.float32, device=b.bbox.device),
                                     b.bbox.float()], 1) for i, b in enumerate(boxes)])

    def _one(self, feat, rois, scale, reduce_mean):
        return ops.roi_align(feat, rois, self.output_size, scale, self.sampling_ratio, aligned=self.aligned,
                             reduce_mean=reduce_mean)

    def forward(self, x, boxes, reduce_mean=False):
        rois = self.convert_to_roi_format(boxes)
        if len(self.scales) == 1:
            return self._one(x[0], rois, self.scales[0], reduce_mean)
        levels = self.map_levels(boxes)
        shape = (len(rois), x[0].shape[1]) + (() if reduce_mean else tuple(self.output_size))
        result = torch.zeros(shape, dtype=torch.float32, device=x[0].device)
        for level, (feat, scale) in enumerate(zip(x, self.scales)):
            idx = torch.nonzero(levels == level).squeeze(1)
            if len(idx):
                result[idx] = self._one(feat, rois[idx], scale, reduce_mean)
        return result


class CustomPooler(Pooler):
    """Features of every box on EVERY FPN level: [L, R, C, P, P] (or [L, R, C] with reduce_mean)."""

    def forward(self, x, boxes, reduce_mean=False):
        rois = self.convert_to_roi_format(boxes)
        if len(self.scales) == 1:
            return self._one(x[0], rois, self.scales[0], reduce_mean)
        return torch.stack([self._one(feat, rois, scale, reduce_mean) for feat, scale in zip(x, self.scales)])
