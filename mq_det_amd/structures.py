"""BoxList / ImageList containers with the reference's semantics (the boundary's input / return types).

Written from the behaviour of maskrcnn_benchmark/structures/bounding_box.py:9-285 (BoxList: bbox [n,4],
size=(w,h), mode, extra fields, legacy +1 width conventions), boxlist_ops.py:148 (cat_boxlist) and
image_list.py:7-70 (ImageList / to_image_list zero-padding to a size divisibility).
"""
import math

import torch


class BoxList:
    def __init__(self, bbox, image_size, mode="xyxy"):
        device = bbox.device if isinstance(bbox, torch.Tensor) else torch.device("cpu")
        bbox = torch.as_tensor(bbox, dtype=torch.float32, device=device)
        if bbox.ndimension() != 2 or bbox.size(-1) != 4:
            raise ValueError(f"bbox should be [n, 4], got {tuple(bbox.shape)}")
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        self.bbox, self.size, self.mode = bbox, tuple(image_size), mode     # size = (width, height)
        self.extra_fields = {}

    # fields
    def add_field(self, name, data):
        self.extra_fields[name] = data

    def get_field(self, name):
        return self.extra_fields[name]

    def has_field(self, name):
        return name in self.extra_fields

    def fields(self):
        return list(self.extra_fields)

    def _like(self, bbox, mode=None):
        out = BoxList(bbox, self.size, mode or self.mode)
        return out

    def convert(self, mode):
        if mode == self.mode:
            return self
        x1, y1, a, b = self.bbox.unbind(-1)
        if mode == "xywh":          # from xyxy, legacy +1
            box = torch.stack([x1, y1, a - x1 + 1, b - y1 + 1], -1)
        else:                       # from xywh
            box = torch.stack([x1, y1, x1 + (a - 1).clamp(min=0), y1 + (b - 1).clamp(min=0)], -1)
        out = self._like(box, mode)
        out.extra_fields = dict(self.extra_fields)
        return out

    def resize(self, size):
        rw, rh = size[0] / self.size[0], size[1] / self.size[1]
        b = self.convert("xyxy").bbox * torch.tensor([rw, rh, rw, rh], device=self.bbox.device)
        out = BoxList(b, size, "xyxy")
        for k, v in self.extra_fields.items():
            out.add_field(k, v.resize(size) if hasattr(v, "resize") and not torch.is_tensor(v) else v)
        return out.convert(self.mode)

    def clip_to_image(self, remove_empty=True):
        w, h = self.size
        b = self.bbox
        self.bbox = torch.stack([b[:, 0].clamp(0, w - 1), b[:, 1].clamp(0, h - 1),
                                 b[:, 2].clamp(0, w - 1), b[:, 3].clamp(0, h - 1)], -1)
        if remove_empty:
            keep = (self.bbox[:, 3] > self.bbox[:, 1]) & (self.bbox[:, 2] > self.bbox[:, 0])
            return self[keep]
        return self

    def area(self):
        b = self.bbox
        if self.mode == "xyxy":
            return (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
        return b[:, 2] * b[:, 3]

    def to(self, device):
        out = BoxList(self.bbox.to(device), self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v.to(device) if hasattr(v, "to") else v)
        return out

    def __getitem__(self, item):
        out = BoxList(self.bbox[item], self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v[item])
        return out

    def __len__(self):
        return self.bbox.shape[0]

    def copy_with_fields(self, fields):
        out = BoxList(self.bbox, self.size, self.mode)
        for f in ([fields] if isinstance(fields, str) else fields):
            out.add_field(f, self.get_field(f))
        return out

    def __repr__(self):
        return f"BoxList(num_boxes={len(self)}, image_width={self.size[0]}, image_height={self.size[1]}, mode={self.mode})"


def cat_boxlist(boxlists):
    assert len(boxlists) > 0 and all(b.size == boxlists[0].size and b.mode == boxlists[0].mode for b in boxlists)
    out = BoxList(torch.cat([b.bbox for b in boxlists], 0), boxlists[0].size, boxlists[0].mode)
    for f in boxlists[0].fields():
        out.add_field(f, torch.cat([b.get_field(f) for b in boxlists], 0))
    return out


class ImageList:
    def __init__(self, tensors, image_sizes):
        self.tensors, self.image_sizes = tensors, image_sizes      # image_sizes: [(h, w)]

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)


def to_image_list(tensors, size_divisible=0):
    if isinstance(tensors, torch.Tensor) and size_divisible > 0:
        tensors = [tensors]
    if isinstance(tensors, ImageList):
        return tensors
    if isinstance(tensors, torch.Tensor):
        assert tensors.dim() == 4
        return ImageList(tensors, [tuple(t.shape[-2:]) for t in tensors])
    if isinstance(tensors, (tuple, list)):
        H = max(t.shape[1] for t in tensors)
        W = max(t.shape[2] for t in tensors)
        if size_divisible > 0:
            H = int(math.ceil(H / size_divisible) * size_divisible)
            W = int(math.ceil(W / size_divisible) * size_divisible)
        batch = tensors[0].new_zeros(len(tensors), tensors[0].shape[0], H, W)
        for img, pad in zip(tensors, batch):
            pad[:, :img.shape[1], :img.shape[2]].copy_(img)
        return ImageList(batch, [tuple(t.shape[-2:]) for t in tensors])
    raise TypeError(f"Unsupported type for to_image_list: {type(tensors)}")
