"""GeneralizedVLRCNN_New -- the drop-in boundary (reference modeling/detector/generalized_vl_rcnn_new.py:90-519,
constructor registry modeling/detector/__init__.py:4-14).

Same constructor (`build_detection_model(cfg)`), same `forward(images, targets=None, captions=None,
positive_map=None, greenlight_map=None, return_backbone_features=False)` -> `list[BoxList]`, same attribute tree
(`backbone.body/.fpn`, `language_backbone`, `rpn.head`, `query_selector`, `tokenizer`) and the same
`state_dict()` keys, so `tools/test_grounding_net.py` / `engine/inference.py` / `GLIPDemo` can drive it
unchanged.  Inference only: training mode raises NotImplementedError.
Differences by design: fp16 compute on MI355X HIP kernels; B > 1 is allowed with vision queries (the
reference asserts B == 1, :354); post-processing has a single device->host sync per forward.
"""
import os

import torch
from torch import nn

from ..structures import BoxList, to_image_list
from . import pipeline
from .params import Container, build_param_tree
from .query_selector import QuerySelector, labels_and_maps, build_token_index


class GeneralizedVLRCNN_New(nn.Module):
    def __init__(self, cfg, tokenizer=None, **kwargs):
        super().__init__()
        self.cfg = cfg
        for name in ("backbone", "language_backbone", "rpn"):
            self.add_module(name, Container())
        build_param_tree(self, cfg)
        self.roi_heads = None                                     # RPN_ONLY (roi_heads/__init__.py:64-84)
        self.query_selector = None if cfg.VISION_QUERY.DISABLE_SELECTOR else QuerySelector(cfg)
        self.tokenizer = tokenizer if tokenizer is not None else self._load_tokenizer(cfg)
        self._plan = None
        self._plan_key = None
        self._anchor_cache = {}
        self.eval()

    @staticmethod
    def _load_tokenizer(cfg):
        from transformers import AutoTokenizer
        name = cfg.MODEL.LANGUAGE_BACKBONE.TOKENIZER_TYPE
        if not os.path.isdir(name):
            raise RuntimeError(
                f"tokenizer files for '{name}' are not on disk (no network here): point "
                "MODEL.LANGUAGE_BACKBONE.TOKENIZER_TYPE at a local directory whose basename is 'bert-base-uncased' "
                "(mq_det_amd.utils.tokenizer.build_synthetic_tokenizer writes one) or pass tokenizer=...")
        return AutoTokenizer.from_pretrained(name)

    # ------------------------------------------------------------------ plan management
    def _invalidate(self):
        self._plan = None
        self._anchor_cache = {}

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._invalidate()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._invalidate()
        return out

    def prepare(self, device=None):
        """(Re)build the fp16 inference plan on `device`.  Called lazily by forward."""
        device = torch.device(device) if device is not None else next(self.parameters()).device
        if device.type != "cuda":
            raise RuntimeError("mq_det_amd runs on MI355X only (HIP kernels, no CPU fallback); got device " + str(device))
        from .. import ops
        ops.load_library()
        self._plan = pipeline.build_plan(self.state_dict(), self.cfg, device)
        self._plan_key = device
        return self._plan

    # ------------------------------------------------------------------ reference API
    def train(self, mode=True):
        if mode:
            raise NotImplementedError("mq_det_amd implements the inference forward only (north-star scope)")
        return super().train(False)

    def load_query_bank(self, query_path):
        self.query_selector.load_query_bank(query_path)

    def extract_query(self, *a, **k):
        raise NotImplementedError("vision-query extraction (ROIAlign pooler) is a 'next' row, SURVEY.md 8f")

    def flatten_fpn_features(self, features):
        return pipeline.pooled_fpn_tokens(features)

    def get_labels_and_maps_from_positive_map(self, positive_map, dtype=torch.float):
        return labels_and_maps(positive_map, self.cfg.MODEL.LANGUAGE_BACKBONE.MAX_QUERY_LEN)

    def tokenize(self, captions, device):
        LB = self.cfg.MODEL.LANGUAGE_BACKBONE
        tok = self.tokenizer(list(captions), max_length=LB.MAX_QUERY_LEN,
                             padding="max_length" if LB.PAD_MAX else "longest",
                             return_special_tokens_mask=True, return_tensors="pt", truncation=True)
        return tok["input_ids"].to(device, non_blocking=True), tok["attention_mask"].to(device, non_blocking=True)

    @torch.no_grad()
    def forward(self, images, targets=None, captions=None, positive_map=None, greenlight_map=None,
                return_backbone_features=False, return_raw=False, input_ids=None, attention_mask=None):
        if self.training:
            raise NotImplementedError("training forward is out of scope")
        images = to_image_list(images)
        dev = images.tensors.device
        if self._plan is None or self._plan_key != dev:
            self.prepare(dev)
        P, cfg = self._plan, self.cfg
        dtype = P["backbone.body.patch_embed.proj.weight"].dtype
        x = images.tensors.to(dtype).contiguous(memory_format=torch.channels_last)
        Bn = x.shape[0]

        feats = pipeline.fpn_forward(P, pipeline.swin_forward(P, cfg, x))

        if input_ids is None:
            input_ids, attention_mask = self.tokenize(captions, dev)
        T = input_ids.shape[1]

        vision = idx = pooled = None
        labels_in_caption = [k for k, v in positive_map.items() if len(v) != 0]
        if cfg.VISION_QUERY.ENABLED and self.query_selector is not None and self.query_selector.query_bank is not None:
            vision, idx = self.query_selector.select([labels_in_caption] * Bn, [positive_map] * Bn, T, dev, dtype)
            pooled = pipeline.pooled_fpn_tokens(feats)
        lang = pipeline.language_backbone(P, cfg, input_ids, attention_mask, vision, pooled, idx,
                                          want_gates=cfg.VISION_QUERY.RETURN_ATTN_GATE_VALUE)
        head = pipeline.vldyhead(P, cfg, feats, lang)

        sizes = tuple(tuple(f.shape[-2:]) for f in feats)
        if sizes not in self._anchor_cache:                       # constant per feature-map geometry
            self._anchor_cache[sizes] = pipeline.grid_anchors(P, sizes, cfg.MODEL.RPN.ANCHOR_STRIDE, dev)
        anchors = self._anchor_cache[sizes]
        tokidx, label_ids = build_token_index(positive_map, labels_in_caption, dev)
        post = pipeline.postprocess(cfg, head, anchors, images.image_sizes, tokidx, label_ids, want_cls=return_raw)
        if return_raw:
            return {"post": post, "head": head, "lang": lang, "feats": feats, "anchors": anchors,
                    "vision": vision, "idx": idx, "pooled": pooled}

        # fixed-shape detections [B, K, 6] for the RCCL all-gather (mq_det_amd.parallel.gather_detections)
        self.last_packed = torch.cat([post["boxes"], post["scores"][..., None], post["labels"].float()[..., None]], -1)
        counts = post["counts"].tolist()                          # the one device->host sync of the forward
        result = []
        for b, (h, w) in enumerate(images.image_sizes):
            n = counts[b]
            bl = BoxList(post["boxes"][b, :n], (int(w), int(h)), mode="xyxy")
            bl.add_field("labels", post["labels"][b, :n])
            bl.add_field("scores", post["scores"][b, :n])
            result.append(bl)
        if cfg.VISION_QUERY.RETURN_ATTN_GATE_VALUE:
            return result, lang["vision_query_gates"]
        if return_backbone_features:
            return result, [f.float().contiguous() for f in feats]
        return result


_DETECTION_META_ARCHITECTURES = {"GeneralizedVLRCNN_New": GeneralizedVLRCNN_New}


def build_detection_model(cfg, **kwargs):
    """modeling/detector/__init__.py:9-14."""
    if cfg.get("GROUNDINGDINO", {}).get("enabled", False):
        raise NotImplementedError("MQ-GroundingDINO (BASELINE configs[4]) is a 'next' row, SURVEY.md 8f")
    return _DETECTION_META_ARCHITECTURES[cfg.MODEL.META_ARCHITECTURE](cfg, **kwargs)
