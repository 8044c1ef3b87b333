"""MQ-GroundingDINO behind the reference's module API, on the MI355X pipeline (gdino_pipeline.py).

Drop-in for `groundingdino_new.models.GroundingDINO.groundingdino.GroundingDINO` (:98-661) as returned by
`build_detection_model(cfg)` when `cfg.GROUNDINGDINO.enabled` (modeling/detector/__init__.py:9-14): same constructor input (the
yacs cfg), same `state_dict()` names (strict `load_state_dict` of a reference checkpoint, incl. the aliased `bbox_embed.*` /
`transformer.decoder.bbox_embed.*` entries of the shared box head), `forward(samples, targets=None, captions=..., positive_map=...)`
-> list[BoxList] (`(result, srcs)` with return_backbone_features), `load_query_bank`, `extract_query`.  Inference only.
Differences by design: vision queries work for any batch size whose images share the caption (the reference asserts B == 1,
:502); a label with an empty token list silences ALL detections exactly like the reference's NaN propagation (:291-305).
"""
import os
from collections import OrderedDict

import torch
from torch import nn

from .. import ops as _ops
from ..structures import BoxList, to_image_list
from . import gdino_pipeline as gp
from .detector import compute_dtype, expand_bbox, pool_into_bank
from .graph_runner import GraphRunner
from .params import build_param_tree, gdino_param_specs, gdino_swin_cfg
from .poolers import CustomPooler, Pooler
from .query_selector import QuerySelector, labels_and_maps


def preprocess_caption(caption):
    """groundingdino.py:92-96."""
    result = caption.lower().strip()
    return result if result.endswith(".") else result + "."


class GroundingDINO(GraphRunner, nn.Module):
    def __init__(self, cfg, tokenizer=None, **kwargs):
        super().__init__()
        self.cfg = cfg
        G = cfg.GROUNDINGDINO
        self._validate_config()
        build_param_tree(self, cfg, specs=gdino_param_specs(cfg))
        self.box_threshold = G.box_threshold
        self.num_queries, self.hidden_dim, self.max_text_len = G.num_queries, G.hidden_dim, 256
        RB = cfg.MODEL.ROI_BOX_HEAD
        pool_cls = Pooler if cfg.VISION_QUERY.SELECT_FPN_LEVEL else CustomPooler
        self.pooler = pool_cls(output_size=(RB.POOLER_RESOLUTION, RB.POOLER_RESOLUTION), scales=RB.POOLER_SCALES,
                               sampling_ratio=RB.POOLER_SAMPLING_RATIO, use_v2=True)
        self.query_selector = None if cfg.VISION_QUERY.DISABLE_SELECTOR else QuerySelector(cfg)
        self.tokenizer = tokenizer if tokenizer is not None else self._load_tokenizer(G.text_encoder_type)
        self.specical_tokens = self.tokenizer.convert_tokens_to_ids(["[CLS]", "[SEP]", ".", "?"])      # (sic) groundingdino.py:194
        self._swin = gdino_swin_cfg(cfg)
        self._plan = self._plan_key = None
        self._kernels = None                                      # kernel selection of the plan (ops.configure), set by prepare()
        self._graphs = OrderedDict()
        self._geo_cache, self._txt_cache, self._map_cache = OrderedDict(), OrderedDict(), OrderedDict()
        self._feat_cache = None                                   # projected levels of the last image batch
        self.backbone_cache = bool(cfg.MODEL.get("BACKBONE_CACHE", True))
        self.use_hip_graph = bool(cfg.MODEL.get("USE_HIP_GRAPH", True))
        self.graph_cache_size = int(cfg.MODEL.get("HIP_GRAPH_CACHE", 8))
        self.graph_warm_calls = int(cfg.MODEL.get("HIP_GRAPH_WARM_CALLS", 1))
        self.cache_stats = {"graph_replay": 0, "graph_capture": 0, "eager": 0, "graph_evict": 0, "backbone_hit": 0, "backbone_miss": 0}
        self.eval()

    @staticmethod
    def _load_tokenizer(name):
        from transformers import AutoTokenizer
        if os.path.basename(name) != "bert-base-uncased":
            raise NotImplementedError("GROUNDINGDINO.text_encoder_type: only bert-base-uncased (groundingdino.py:183-184)")
        if not os.path.isdir(name):
            raise RuntimeError(f"tokenizer files for '{name}' are not on disk (no network here): point GROUNDINGDINO.text_encoder_type "
                               "at a local directory whose basename is 'bert-base-uncased' or pass tokenizer=...")
        return AutoTokenizer.from_pretrained(name)

    def _validate_config(self):
        G = self.cfg.GROUNDINGDINO
        want = dict(two_stage_type="standard", embed_init_tgt=True, use_text_enhancer=True, use_fusion_layer=True,
                    use_text_cross_attention=True, sub_sentence_present=True, dec_pred_bbox_embed_share=True, num_patterns=0,
                    query_dim=4, num_feature_levels=4, enc_n_points=4, dec_n_points=4, hidden_dim=256, nheads=8,
                    transformer_activation="relu", pre_norm=False, position_embedding="sine", max_text_len=256)
        for k, v in want.items():
            if G.get(k, v) != v:
                raise NotImplementedError(f"GROUNDINGDINO.{k} = {G[k]}: this path implements {v!r} (configs/pretrain/mq-groundingdino-t.yaml)")
        if list(G.return_interm_indices) != [1, 2, 3]:
            raise NotImplementedError("GROUNDINGDINO.return_interm_indices must be [1, 2, 3]")
        if G.pe_temperatureH != G.pe_temperatureW:
            raise NotImplementedError("GROUNDINGDINO.pe_temperatureH != pe_temperatureW")
        V = self.cfg.VISION_QUERY
        if V.get("ADD_ADAPT_LAYER", False) or V.get("QUERY_FUSION", False) or V.get("LEARNABLE_BANK", False):
            raise NotImplementedError("VISION_QUERY.ADD_ADAPT_LAYER / QUERY_FUSION / LEARNABLE_BANK are not implemented")

    # ------------------------------------------------------------------ plan management
    def _invalidate(self):
        self._plan = None
        self._drop_graphs()
        self._geo_cache = OrderedDict()
        self._feat_cache = None

    def clear_caches(self):
        self._feat_cache = None
        self._drop_graphs()

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._invalidate()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._invalidate()
        return out

    def prepare(self, device=None):
        device = torch.device(device) if device is not None else next(self.parameters()).device
        if device.type != "cuda":
            raise RuntimeError("mq_det_amd runs on MI355X only (HIP kernels, no CPU fallback); got device " + str(device))
        from .. import ops
        ops.load_library()
        self._kernels = dict(ops.configure(self.cfg))              # kernel selection: read once per plan, kept WITH the plan
        self._plan = gp.build_gdino_plan(self.state_dict(), self.cfg, device, self._swin, dtype=compute_dtype(self.cfg))
        self._plan_key = device
        return self._plan

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("mq_det_amd implements the inference forward only (north-star scope)")
        return super().train(False)

    def load_query_bank(self, query_path):
        self.query_selector.load_query_bank(query_path)

    def _use_vq(self):
        return bool(self.cfg.VISION_QUERY.ENABLED and self.query_selector is not None
                    and self.query_selector.query_bank is not None)

    def flatten_fpn_features(self, features):
        from . import pipeline
        return pipeline.pooled_fpn_tokens(features)

    def get_labels_and_maps_from_positive_map(self, positive_map, dtype=torch.float):
        return labels_and_maps(positive_map, self.cfg.MODEL.LANGUAGE_BACKBONE.MAX_QUERY_LEN)

    # ------------------------------------------------------------------ memoised host-side glue
    @staticmethod
    def _memo(cache, key, make, cap=64):
        hit = cache.get(key)
        if hit is None:
            hit = cache[key] = make()
            while len(cache) > cap:
                cache.popitem(last=False)
        else:
            cache.move_to_end(key)
        return hit

    def _text(self, captions, dev):
        def make():
            tok = self.tokenizer(list(captions), padding="max_length", return_tensors="pt")          # groundingdino.py:518
            return gp.text_inputs(self.cfg, tok["input_ids"], tok["attention_mask"], self.specical_tokens, dev)
        return self._memo(self._txt_cache, (tuple(captions), str(dev)), make)

    def _class_map(self, positive_map, dev):
        """[T, C] fp32: column label-1 holds 1/len over the label's tokens (convert_grounding_to_od_logits, MEAN);
        returns (map, has_empty_label)."""
        T, C = self.max_text_len, self.cfg.MODEL.DYHEAD.NUM_CLASSES - 1
        key = (tuple((k, (v,) if isinstance(v, int) else tuple(v)) for k, v in positive_map.items()), str(dev))

        def make():
            m = torch.zeros(T, C)
            empty = False
            for lab, toks in positive_map.items():
                toks = [toks] if isinstance(toks, int) else list(toks)
                if not toks:
                    empty = True
                    continue
                m[:, lab - 1] = 0.0
                for t in toks:
                    m[t, lab - 1] += 1.0 / len(toks)
            return m.to(dev), empty
        return self._memo(self._map_cache, key, make)

    # ------------------------------------------------------------------ device program (capturable)
    def _program(self, x, geo, txt, vision, idx, class_map, im_hw, nan_labels, max_kv=0, trace=None):
        return gp.forward_device(self._plan, self.cfg, self._swin, x, geo, txt, vision, idx, class_map, im_hw, max_kv=int(max_kv),
                                 nan_labels=bool(nan_labels), trace=trace)

    def _program_rest(self, src32, geo, txt, vision, idx, class_map, im_hw, nan_labels, max_kv=0):
        """From the cached projected levels of the same pixels (Swin + input projections skipped)."""
        return gp.forward_device(self._plan, self.cfg, self._swin, None, geo, txt, vision, idx, class_map, im_hw, max_kv=int(max_kv),
                                 nan_labels=bool(nan_labels), src32=src32)

    @torch.no_grad()
    def forward(self, samples, targets=None, return_raw=False, **kw):
        if self.training:
            raise NotImplementedError("training forward is out of scope")
        if targets is not None:
            captions = [t.get_field("caption") for t in targets if "caption" in t.fields()]
        else:
            captions = kw["captions"]
        captions = [preprocess_caption(c) for c in captions]
        positive_map = kw["positive_map"]
        # token positions cut away by the truncation to max_text_len cannot be scored (the reference would index past the
        # [.., 256] token scores): dropped, like the MQ-GLIP class does
        T = self.max_text_len
        if any(t >= T for v in positive_map.values() for t in ([v] if isinstance(v, int) else v)):
            positive_map = {k: [t for t in ([v] if isinstance(v, int) else v) if t < T] for k, v in positive_map.items()}
        return_backbone_features = kw.get("return_backbone_features", False)
        images = to_image_list(samples)
        dev = images.tensors.device
        if self._plan is None or self._plan_key != dev:
            self.prepare(dev)
        _ops.activate(self._kernels)
        P = self._plan
        dtype = P["backbone.0.patch_embed.proj.weight"].dtype
        Bn, _, H, W = images.tensors.shape
        sizes = tuple((int(h), int(w)) for h, w in images.image_sizes)
        geo = self._memo(self._geo_cache, (H, W, sizes, str(dev)), lambda: gp.geometry(P, self.cfg, H, W, sizes, dev), cap=16)
        txt, max_kv = self._text(captions, dev)
        class_map, nan_labels = self._class_map(positive_map, dev)
        vision = idx = None
        if self._use_vq():
            T = self.cfg.MODEL.LANGUAGE_BACKBONE.MAX_QUERY_LEN
            labels_in_caption = [k for k, v in positive_map.items() if len(v) != 0]
            pm_key = tuple((k, tuple(positive_map[k])) for k in labels_in_caption)
            vision, idx = self.query_selector.select_cached(pm_key, labels_in_caption, positive_map, Bn, T, dev, dtype)
            if vision.shape[1] == 0:
                vision = idx = None
        im_hw = torch.tensor([[h, w] for (h, w) in sizes], dtype=torch.float32, device=dev)
        x = images.tensors.to(dtype).contiguous(memory_format=torch.channels_last)
        inputs = (x, geo, txt, vision, idx, class_map, im_hw, nan_labels, max_kv)
        if return_raw:
            trace = {}
            out = self._program(*inputs, trace=trace)
            trace.update(out=out, geo=geo, txt=txt)
            return trace
        from .. import ops
        use_graph = self.use_hip_graph and not ops.timing_active()
        # f1: the pixels of the previous call (same tensor object, not modified since) -> cached projected levels; the strong
        # reference to the input tensor keeps its storage alive, so identity + version counter cannot alias another batch
        # (a writer that refills `images.tensors` without bumping its version counter must pass reuse_backbone=False -- see
        # GeneralizedVLRCNN_New.forward)
        fc, src = (self._feat_cache if (self.backbone_cache and kw.get("reuse_backbone") is not False) else None), images.tensors
        if fc is not None and fc["src"] is src and fc["version"] == src._version:
            self.cache_stats["backbone_hit"] += 1
            out = self._run("_program_rest", (fc["src32"],) + inputs[1:], use_graph)
        else:
            out = self._run("_program", inputs, use_graph)
            if self.backbone_cache:
                self.cache_stats["backbone_miss"] += 1
                self._feat_cache = {"src": src, "version": src._version, "src32": out["srcs"].clone()}
        self.last_packed = packed = out["packed"].clone()
        nz = out["keep"].nonzero()                                 # [n, 2] (image, query) in query order: the one device -> host
        counts = torch.bincount(nz[:, 0], minlength=Bn).tolist()   # sync of the forward
        sel = packed[nz[:, 0], nz[:, 1]]
        result, s0 = [], 0
        for b, (h, w) in enumerate(sizes):
            part = sel[s0:s0 + counts[b]]
            s0 += counts[b]
            bl = BoxList(part[:, :4].clone(), (int(w), int(h)), mode="xyxy")
            bl.add_field("labels", part[:, 5].to(torch.int64))
            bl.add_field("scores", part[:, 4].clone())
            result.append(bl)
        if return_backbone_features:
            s0, feats = 0, []
            for (h, w) in geo["shapes"]:                           # `srcs` of the reference: the projected levels, NCHW
                feats.append(out["srcs"][:, s0:s0 + h * w].reshape(Bn, h, w, -1).permute(0, 3, 1, 2).clone())
                s0 += h * w
            return result, feats
        return result

    @torch.no_grad()
    def extract_query(self, samples=None, targets=None, query_images=None, visual_features=None, exclude_similar=False,
                      device=None, max_query_number=None):
        """groundingdino.py:340-421: ROI-pool the (projected) feature levels under the expanded target boxes into the bank."""
        cfg = self.cfg
        device = torch.device(device) if device else (to_image_list(samples).tensors.device if samples is not None
                                                      else visual_features[0].device)
        targets = expand_bbox([t.to(device) for t in targets if t is not None], expand_ratio=cfg.VISION_QUERY.EXPAND_RATIO)
        if visual_features is None:
            images = to_image_list(samples)
            if self._plan is None or self._plan_key != images.tensors.device:
                self.prepare(images.tensors.device)
            _ops.activate(self._kernels)
            from . import pipeline
            P = self._plan
            x = images.tensors.to(P["backbone.0.patch_embed.proj.weight"].dtype).contiguous(memory_format=torch.channels_last)
            src = gp.input_projections(P, cfg, pipeline.swin_forward(P, cfg, x, p="backbone.0", SW=self._swin))
            visual_features, s0 = [], 0
            for (h, w) in gp.level_shapes(x.shape[2], x.shape[3], cfg.GROUNDINGDINO.num_feature_levels):
                visual_features.append(src[:, s0:s0 + h * w].reshape(x.shape[0], h, w, -1).permute(0, 3, 1, 2))
                s0 += h * w
        else:
            visual_features = [v.to(device) for v in visual_features]
        return pool_into_bank(cfg, self.pooler, visual_features, targets, query_images, exclude_similar, max_query_number)
