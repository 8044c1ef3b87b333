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
        self._graphs, self._tok_cache, self._tokidx_cache, self._wh_cache = {}, {}, {}, {}
        self.use_hip_graph = bool(cfg.MODEL.get("USE_HIP_GRAPH", True))
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
        self._graphs = {}

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
        """HF tokenizer on the host (generalized_vl_rcnn_new.py:378-383); memoised per caption tuple -- the LVIS
        protocol re-sends the same 31 chunk captions for every image (engine/inference.py:605-625)."""
        key = (tuple(captions), str(device))
        hit = self._tok_cache.get(key)
        if hit is None:
            LB = self.cfg.MODEL.LANGUAGE_BACKBONE
            tok = self.tokenizer(list(captions), max_length=LB.MAX_QUERY_LEN,
                                 padding="max_length" if LB.PAD_MAX else "longest",
                                 return_special_tokens_mask=True, return_tensors="pt", truncation=True)
            am = tok["attention_mask"]
            # host-side bound of the per-caption key length (last attended position + 1): picks the kernel variant of
            # the VLFuse image-side attention, and is part of the HIP-graph key
            max_kv = int((am * torch.arange(1, am.shape[1] + 1)).max())
            hit = (tok["input_ids"].to(device), am.to(device), max_kv)
            if len(self._tok_cache) > 256:
                self._tok_cache.clear()
            self._tok_cache[key] = hit
        return hit

    # ------------------------------------------------------------------ device part (capturable in a HIP graph)
    def _device_forward(self, x, input_ids, attention_mask, vision, idx, tokidx, label_ids, im_wh, max_kv=0, want_raw=False):
        P, cfg = self._plan, self.cfg
        # the image-independent part of the language backbone (embeddings + BERT layers below the first GCP block) runs on
        # a side stream under the Swin backbone: its launches are tiny (B x 256 tokens) and would otherwise serialise
        front = None
        if x.is_cuda and cfg.MODEL.DYHEAD.get("LEVEL_STREAMS", True):
            main, text = torch.cuda.current_stream(), pipeline._side_streams(x.device, 1, "text")[0]
            text.wait_stream(main)
            with torch.cuda.stream(text):
                front = pipeline.language_front(P, cfg, input_ids, attention_mask, vision is not None)
        feats = pipeline.fpn_forward(P, pipeline.swin_forward(P, cfg, x))
        if front is not None:
            main.wait_stream(text)
        pooled = pipeline.pooled_fpn_tokens(feats) if vision is not None else None
        lang = pipeline.language_backbone(P, cfg, input_ids, attention_mask, vision, pooled, idx,
                                          want_gates=cfg.VISION_QUERY.RETURN_ATTN_GATE_VALUE, front=front)
        lang["max_kv"] = max_kv
        head = pipeline.vldyhead(P, cfg, feats, lang)
        sizes = tuple(tuple(f.shape[-2:]) for f in feats)
        if sizes not in self._anchor_cache:                       # constant per feature-map geometry
            self._anchor_cache[sizes] = pipeline.grid_anchors(P, sizes, cfg.MODEL.RPN.ANCHOR_STRIDE, x.device)
        anchors = self._anchor_cache[sizes]
        post = pipeline.postprocess(cfg, head, anchors, im_wh, tokidx, label_ids, want_cls=want_raw)
        if want_raw:
            return {"post": post, "head": head, "lang": lang, "feats": feats, "anchors": anchors,
                    "vision": vision, "idx": idx, "pooled": pooled}
        packed = torch.cat([post["boxes"], post["scores"][..., None], post["labels"].float()[..., None]], -1)
        return {"packed": packed, "counts": post["counts"], "feats": feats, "gates": lang["vision_query_gates"]}

    def _graph_forward(self, key, inputs):
        """Replay the whole device forward as ONE HIP graph (static shapes per key).  The eager forward issues
        ~1500 launches per step and was host-bound by ~17 ms / step (profiles/r01_call3); a replay costs one launch.
        First call with a key runs eagerly (library autotuning, caches), the second captures."""
        ent = self._graphs.get(key)
        if ent is None:
            self._graphs[key] = {"stage": 1}
            return self._device_forward(*inputs)
        if ent["stage"] == 1:
            static_in = [t.clone() if torch.is_tensor(t) else t for t in inputs]
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            try:
                # thread_local: with torch.distributed initialised, the RCCL watchdog thread polls events concurrently;
                # under the default "global" mode that would invalidate the capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    static_out = self._device_forward(*static_in)
            except Exception as e:                              # keep running eagerly, but say so loudly
                import warnings
                warnings.warn(f"mq_det_amd: HIP graph capture failed ({type(e).__name__}: {e}); staying eager")
                self._graphs[key] = {"stage": -1}
                torch.cuda.synchronize()
                return self._device_forward(*inputs)
            ent.update(stage=2, graph=g, inp=static_in, out=static_out)
        if ent["stage"] == -1:
            return self._device_forward(*inputs)
        for dst, src in zip(ent["inp"], inputs):
            if torch.is_tensor(dst):
                dst.copy_(src, non_blocking=True)
        ent["graph"].replay()
        return ent["out"]

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
        if input_ids is None:
            input_ids, attention_mask, max_kv = self.tokenize(captions, dev)
        else:                                                     # caller-supplied ids (tests): one host sync
            max_kv = int((attention_mask.cpu() * torch.arange(1, attention_mask.shape[1] + 1)).max())
        T = input_ids.shape[1]

        # host-side glue: all memoised, no device sync
        labels_in_caption = [k for k, v in positive_map.items() if len(v) != 0]
        pm_key = tuple((k, tuple(positive_map[k])) for k in labels_in_caption)
        vision = idx = None
        if cfg.VISION_QUERY.ENABLED and self.query_selector is not None and self.query_selector.query_bank is not None:
            vision, idx = self.query_selector.select_cached(pm_key, labels_in_caption, positive_map, Bn, T, dev, dtype)
        hit = self._tokidx_cache.get((pm_key, str(dev)))
        if hit is None:
            hit = self._tokidx_cache[(pm_key, str(dev))] = build_token_index(positive_map, labels_in_caption, dev)
        tokidx, label_ids = hit
        wh_key = (tuple(images.image_sizes), str(dev))
        im_wh = self._wh_cache.get(wh_key)
        if im_wh is None:
            im_wh = self._wh_cache[wh_key] = torch.tensor([[w, h] for (h, w) in images.image_sizes],
                                                          dtype=torch.float32, device=dev)
        inputs = (x, input_ids, attention_mask, vision, idx, tokidx, label_ids, im_wh, max_kv)
        if return_raw:
            return self._device_forward(*inputs, want_raw=True)
        from .. import ops
        use_graph = self.use_hip_graph and not ops.timing_active()
        if use_graph:
            key = (tuple(x.shape), T, None if vision is None else tuple(vision.shape), None if idx is None else tuple(idx.shape),
                   tuple(tokidx.shape), wh_key, -(-max_kv // 64))
            out = self._graph_forward(key, inputs)
        else:
            out = self._device_forward(*inputs)

        # fixed-shape detections [B, K, 6] for the RCCL all-gather (mq_det_amd.parallel.gather_detections)
        self.last_packed = packed = out["packed"]
        counts = out["counts"].tolist()                           # the one device->host sync of the forward
        result = []
        for b, (h, w) in enumerate(images.image_sizes):
            n = counts[b]
            bl = BoxList(packed[b, :n, :4].clone(), (int(w), int(h)), mode="xyxy")
            bl.add_field("labels", packed[b, :n, 5].to(torch.int64))
            bl.add_field("scores", packed[b, :n, 4].clone())
            result.append(bl)
        if cfg.VISION_QUERY.RETURN_ATTN_GATE_VALUE:
            return result, out["gates"]
        if return_backbone_features:
            return result, [f.float().contiguous() for f in out["feats"]]
        return result


_DETECTION_META_ARCHITECTURES = {"GeneralizedVLRCNN_New": GeneralizedVLRCNN_New}


def build_detection_model(cfg, **kwargs):
    """modeling/detector/__init__.py:9-14."""
    if cfg.get("GROUNDINGDINO", {}).get("enabled", False):
        raise NotImplementedError("MQ-GroundingDINO (BASELINE configs[4]) is a 'next' row, SURVEY.md 8f")
    return _DETECTION_META_ARCHITECTURES[cfg.MODEL.META_ARCHITECTURE](cfg, **kwargs)
