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
from collections import OrderedDict

import torch
from torch import nn

from .. import ops as _ops
from ..structures import BoxList, to_image_list
from . import pipeline
from .poolers import CustomPooler, Pooler
from .graph_runner import GraphRunner, memoised
from .params import Container, build_param_tree
from .query_selector import QuerySelector, labels_and_maps, build_token_index


def expand_bbox(box_list, expand_ratio=1.5):
    """generalized_vl_rcnn_new.py:32-49: grow every box about its centre, clip to the image, drop empty boxes."""
    out = []
    for boxes in box_list:
        assert boxes.mode == "xyxy"
        bbox = boxes.bbox
        w, h = bbox[:, 2] - bbox[:, 0], bbox[:, 3] - bbox[:, 1]
        dw, dh = (w * expand_ratio - w) / 2, (h * expand_ratio - h) / 2
        nb = BoxList(bbox + torch.stack([-dw, -dh, dw, dh], dim=1), boxes.size, mode="xyxy")
        nb.add_field("labels", boxes.get_field("labels"))
        out.append(nb.clip_to_image(remove_empty=True))
    return out


def compute_dtype(cfg):
    """MODEL.COMPUTE_DTYPE: the operand type of every kernel on the path -- "float16" (default, BASELINE.json configs[1]),
    "bfloat16" (configs[3]: the *_bf16 entry points of include/mqdet_hip.h) or "float32" (the precise mode: the *_f32 entry points, the
    same kernel sources with fp32 operands, and fp32 library GEMMs -- a quarter of the MFMA rate, for parity at the north-star's 1e-3 end
    to end, not for throughput); accumulation and residual streams are fp32 in every mode."""
    name = str(cfg.MODEL.get("COMPUTE_DTYPE", "float16")).lower()
    if name in ("float16", "fp16", "half"):
        return torch.float16
    if name in ("bfloat16", "bf16"):
        return torch.bfloat16
    if name in ("float32", "fp32", "float"):
        return torch.float32
    raise NotImplementedError(f"MODEL.COMPUTE_DTYPE = {name}: float16, bfloat16 or float32")


def pool_into_bank(cfg, pooler, visual_features, targets, query_images, exclude_similar, max_query_number):
    """Shared tail of `extract_query` (generalized_vl_rcnn_new.py:264-288 == groundingdino.py:397-421): ROI-pool the (already
    expanded) target boxes -- from their own FPN level (SELECT_FPN_LEVEL) or from all five --, average the bins inside the kernel,
    and append the features to the bank of their label up to `max_query_number`, optionally skipping near-duplicates."""
    query_feats = pooler(visual_features, targets, reduce_mean=True)                # [boxes, C] or [scales, boxes, C]
    if cfg.VISION_QUERY.SELECT_FPN_LEVEL:
        query_feats = query_feats[None]
    else:
        assert len(visual_features) == len(query_feats) == 5
    query_feats = query_feats.permute(1, 0, 2)                                      # boxes, scales, channels
    labels = torch.cat([t.get_field("labels") for t in targets])
    assert len(labels) == len(query_feats)
    max_query_number = cfg.VISION_QUERY.MAX_QUERY_NUMBER if max_query_number is None else max_query_number
    thr = cfg.VISION_QUERY.SIMILARITY_THRESHOLD
    for label, feat in zip(labels.tolist(), query_feats):
        cur = query_images[label] if (label in query_images or hasattr(query_images, "default_factory")) else []
        n = len(cur)
        if n >= max_query_number:
            continue
        if exclude_similar and n > 0:
            assert feat.shape[0] == 1
            bank = torch.nn.functional.normalize(cur.to(feat), p=2, dim=-1)
            new = torch.nn.functional.normalize(feat, p=2, dim=-1)
            if (torch.einsum("bnd,nd->bn", bank, new) > thr).sum() > 0:
                continue
        query_images[label] = feat[None] if n == 0 else torch.cat([cur.to(feat), feat[None]])
    return query_images


class GeneralizedVLRCNN_New(GraphRunner, nn.Module):
    def __init__(self, cfg, tokenizer=None, **kwargs):
        super().__init__()
        self.cfg = cfg
        for name in ("backbone", "language_backbone", "rpn"):
            self.add_module(name, Container())
        build_param_tree(self, cfg)
        self.roi_heads = None                                     # RPN_ONLY (roi_heads/__init__.py:64-84)
        # box pooler of the query-extraction path (generalized_vl_rcnn_new.py:107-121)
        RB = cfg.MODEL.ROI_BOX_HEAD
        pool_cls = Pooler if cfg.VISION_QUERY.SELECT_FPN_LEVEL else CustomPooler
        self.pooler = pool_cls(output_size=(RB.POOLER_RESOLUTION, RB.POOLER_RESOLUTION), scales=RB.POOLER_SCALES,
                               sampling_ratio=RB.POOLER_SAMPLING_RATIO, use_v2=True)
        self.query_selector = None if cfg.VISION_QUERY.DISABLE_SELECTOR else QuerySelector(cfg)
        self.tokenizer = tokenizer if tokenizer is not None else self._load_tokenizer(cfg)
        self._plan = None
        self._plan_key = None
        self._kernels = None                                      # kernel selection of the plan (ops.configure), set by prepare()
        self._anchor_cache = {}
        self._graphs = OrderedDict()                              # LRU of captured HIP graphs, keyed by static shapes only
        self._graph_pool = None                                   # one memory pool shared by every captured graph
        self._tok_cache, self._tokidx_cache, self._wh_cache, self._live_cache = {}, {}, {}, {}
        self._feat_cache = None                                   # Swin + FPN + pooled tokens of the last image batch (f1)
        self._front_cache = OrderedDict()                         # image-independent BERT layers per caption (f1)
        self.use_hip_graph = bool(cfg.MODEL.get("USE_HIP_GRAPH", True))
        # lanes of the staggered schedule (_staggered_program); 1 = the whole batch as one lane.  MQ_MICRO_BATCHES overrides (A/B runs)
        self.micro_batches = int(os.environ.get("MQ_MICRO_BATCHES", cfg.MODEL.get("MICRO_BATCHES", 1)))
        self.graph_cache_size = int(cfg.MODEL.get("HIP_GRAPH_CACHE", 8))
        self.graph_warm_calls = int(cfg.MODEL.get("HIP_GRAPH_WARM_CALLS", 1))
        self.backbone_cache = bool(cfg.MODEL.get("BACKBONE_CACHE", True))
        self.cache_stats = {"backbone_hit": 0, "backbone_miss": 0, "front_hit": 0, "front_miss": 0,
                            "graph_replay": 0, "graph_capture": 0, "eager": 0, "graph_evict": 0}
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
        self._drop_graphs()
        self._feat_cache = None
        self._front_cache = OrderedDict()

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
        self._kernels = dict(ops.configure(self.cfg))              # kernel selection: read once per plan, kept WITH the plan
        self._validate_config()
        self._plan = pipeline.build_plan(self.state_dict(), self.cfg, device, dtype=compute_dtype(self.cfg))
        self._plan_key = device
        return self._plan

    def _validate_config(self):
        """Fail early, with the config key named, on settings of the reference this path does not implement (instead of a
        negative return code from a kernel in the middle of a forward)."""
        cfg = self.cfg
        M = cfg.MODEL
        agg = str(M.DYHEAD.get("SCORE_AGG", "MEAN")).upper()
        mdetr = cfg.TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM != -1
        if agg not in ("MEAN", "MAX", "ONEHOT") + (("POWER",) if mdetr else ()):
            # the reference's own rule: convert_grounding_to_od_logits (rpn/inference.py:772-792) knows MEAN / MAX / ONEHOT,
            # only the MDETR-style _v2 (:795-824) adds POWER; anything else raises NotImplementedError there too
            raise NotImplementedError(f"MODEL.DYHEAD.SCORE_AGG = {agg}: MEAN / MAX / ONEHOT"
                                      + (" / POWER" if mdetr else " (POWER only with TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM != -1)"))
        if M.LANGUAGE_BACKBONE.MAX_QUERY_LEN > 256 or M.LANGUAGE_BACKBONE.MAX_QUERY_LEN % 8:
            raise NotImplementedError("MODEL.LANGUAGE_BACKBONE.MAX_QUERY_LEN must be a multiple of 8 and <= 256 (VLFuse kernels)")
        if M.BACKBONE.OUT_CHANNELS != 256 or M.DYHEAD.CHANNELS != 256:
            raise NotImplementedError("the VLDyHead kernels (VLFuse, DCNv2, DyConv epilogue) are written for 256 channels")
        if M.SWINT.WINDOW_SIZE ** 2 > 160:
            raise NotImplementedError("MODEL.SWINT.WINDOW_SIZE: windows of up to 160 tokens (7x7, 12x12) are supported")
        fc = M.DYHEAD.FUSE_CONFIG
        if fc.TYPE != "MHA-B" or fc.get("SEPARATE_BIDIRECTIONAL", False):
            raise NotImplementedError("MODEL.DYHEAD.FUSE_CONFIG: only TYPE = MHA-B without SEPARATE_BIDIRECTIONAL (mq-glip-*.yaml)")
        if cfg.VISION_QUERY.get("ADD_ADAPT_LAYER", False) or cfg.VISION_QUERY.get("QUERY_FUSION", False) or \
                cfg.VISION_QUERY.get("AUGMENT_IMAGE_WITH_QUERY", False):
            raise NotImplementedError("VISION_QUERY.ADD_ADAPT_LAYER / QUERY_FUSION / AUGMENT_IMAGE_WITH_QUERY are not implemented")

    # ------------------------------------------------------------------ reference API
    def train(self, mode=True):
        if mode:
            raise NotImplementedError("mq_det_amd implements the inference forward only (north-star scope)")
        return super().train(False)

    def load_query_bank(self, query_path):
        self.query_selector.load_query_bank(query_path)

    @torch.no_grad()
    def extract_query(self, images=None, targets=None, query_images=None, visual_features=None, exclude_similar=False,
                      device=None, max_query_number=None):
        """Vision-query extraction (generalized_vl_rcnn_new.py:232-288; callers: tools/extract_vision_query.py via
        tools/train_net.py:303, engine/inference.py:474,492 `online_update`).  targets: list[BoxList] (xyxy, field
        "labels") per image; query_images: the bank being built, `{label: Tensor[n, scales, C]}` (a `defaultdict(list)` in
        the reference); visual_features: FPN maps from `forward(..., return_backbone_features=True)`, or None -> Swin + FPN run
        here on `images`.  Boxes are grown by VISION_QUERY.EXPAND_RATIO, pooled with the aligned ROIAlign (HIP kernel, the
        7 x 7 bins averaged inside the kernel) from their FPN level (SELECT_FPN_LEVEL) or from all five, and appended to
        the bank of their label up to `max_query_number`, optionally skipping near-duplicates (cosine > SIMILARITY_THRESHOLD)."""
        cfg = self.cfg
        device = torch.device(device) if device else (images.tensors.device if images is not None else visual_features[0].device)
        targets = expand_bbox([t.to(device) for t in targets if t is not None], expand_ratio=cfg.VISION_QUERY.EXPAND_RATIO)
        if visual_features is None:
            images = to_image_list(images)
            if self._plan is None or self._plan_key != images.tensors.device:
                self.prepare(images.tensors.device)
            _ops.activate(self._kernels)
            dtype = self._plan["backbone.body.patch_embed.proj.weight"].dtype
            x = images.tensors.to(dtype).contiguous(memory_format=torch.channels_last)
            visual_features, _ = self._backbone_stage(x)
        else:
            visual_features = [v.to(device) for v in visual_features]
        return pool_into_bank(cfg, self.pooler, visual_features, targets, query_images, exclude_similar, max_query_number)

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
            # PAD_MAX = False ("longest" in the reference, generalized_vl_rcnn_new.py:378-383) pads to MAX_QUERY_LEN here as well:
            # padded positions are masked keys everywhere (BERT, GCP, VLFuse) and never scored, so the detections are the same;
            # forward() cuts the padded ROWS away again (`_live_len`: the first 16 ceil(max_kv / 16) positions go to the device programs)
            tok = self.tokenizer(list(captions), max_length=LB.MAX_QUERY_LEN, padding="max_length",
                                 return_special_tokens_mask=True, return_tensors="pt", truncation=True)
            am = tok["attention_mask"]
            # host-side bound of the per-caption key length (last attended position + 1): picks the kernel variant of
            # the VLFuse image-side attention, and is part of the HIP-graph key
            max_kv = int((am * torch.arange(1, am.shape[1] + 1)).max())
            hit = (memoised(tok["input_ids"].to(device)), memoised(am.to(device)), max_kv)
            if len(self._tok_cache) > 256:
                self._tok_cache.clear()
            self._tok_cache[key] = hit
        return hit

    def _live_len(self, T, max_kv):
        """Text positions the device programs run on: 16 ceil(max_kv / 16) (the key-block granularity of the attention kernels; VLFuse takes
        T % 8 == 0), or all T when the bound is unknown or MODEL.LANGUAGE_BACKBONE.COMPACT_TEXT is off."""
        if max_kv <= 0 or not self.cfg.MODEL.LANGUAGE_BACKBONE.get("COMPACT_TEXT", True):
            return T
        if self.cfg.VISION_QUERY.get("RETURN_ATTN_GATE_VALUE", False):
            # the reference's attn_gate.mean() averages over all B x MAX_QUERY_LEN positions (padded rows contribute the constant
            # tanh(gate(LN(0)))): the diagnostic value is only the reference's when every padded row is computed -- no compaction with it
            return T
        return min(T, -(-int(max_kv) // 16) * 16)

    def _live_slice(self, input_ids, attention_mask, Tl, cap_key, dev):
        """input_ids / attention_mask cut to their first Tl columns as contiguous tensors (memoised per caption: the HIP-graph replay copies
        its inputs into static buffers, the same two tensors serve every call of a caption)."""
        key = (cap_key, Tl, str(dev)) if cap_key is not None else None
        hit = self._live_cache.get(key) if key is not None else None
        if hit is None:
            hit = (input_ids[:, :Tl].contiguous(), attention_mask[:, :Tl].contiguous())
            if key is not None:
                memoised(hit)
                if len(self._live_cache) > 256:
                    self._live_cache.clear()
                self._live_cache[key] = hit
        return hit

    @staticmethod
    def _pad_raw_text(raw, T):
        """return_raw (parity ladders): the per-token tensors of a compacted forward back in the caller's [.., T, ..] shapes -- zeros at the
        padded positions (which the reference fills with values nothing reads)."""
        import torch.nn.functional as F

        def rows(t_):            # [B, Tl, C] -> [B, T, C]
            return t_ if t_ is None or t_.shape[1] >= T else F.pad(t_, (0, 0, 0, T - t_.shape[1]))

        def cols(t_):            # [.., Tl] -> [.., T]
            return t_ if t_ is None or t_.shape[-1] >= T else F.pad(t_, (0, T - t_.shape[-1]))
        lang, head = raw["lang"], raw["head"]
        for k in ("hidden", "hidden32", "embedded"):
            if torch.is_tensor(lang.get(k)):
                lang[k] = rows(lang[k])
        for k in ("masks", "key_bias"):
            if torch.is_tensor(lang.get(k)):
                lang[k] = cols(lang[k])
        if torch.is_tensor(head.get("hidden")):
            head["hidden"] = rows(head["hidden"])
        # (head["tbias"], head["dot"] and the alignment operands stay Tl wide: postprocess() may be called on this head again; their first
        # max_kv columns are what the ladders read)
        for a in raw.get("head_trace") or ():
            if torch.is_tensor(a.get("bert_hidden")):
                a["bert_hidden"] = rows(a["bert_hidden"])
        return raw

    # ------------------------------------------------------------------ device part (capturable in a HIP graph)
    def _backbone_stage(self, x):
        """Swin + FPN (+ the pooled FPN tokens the GCP pre-select attends to): depends on the pixels only."""
        P, cfg = self._plan, self.cfg
        feats = pipeline.fpn_forward(P, pipeline.swin_forward(P, cfg, x))
        pooled = pipeline.pooled_fpn_tokens(feats) if self._use_vq() else None
        return feats, pooled

    def _use_vq(self):
        return bool(self.cfg.VISION_QUERY.ENABLED and self.query_selector is not None
                    and self.query_selector.query_bank is not None)

    def _head_stage(self, feats, pooled, front, input_ids, attention_mask, vision, idx, tokidx, label_ids, im_wh, max_kv,
                    want_raw=False):
        """Image-dependent half of the language backbone (pre-select + GCP / BERT layers), VLDyHead, post-processing."""
        P, cfg = self._plan, self.cfg
        lang = pipeline.language_backbone(P, cfg, input_ids, attention_mask, vision, pooled, idx,
                                          want_gates=cfg.VISION_QUERY.RETURN_ATTN_GATE_VALUE, front=front, max_kv=max_kv,
                                          side_ok=bool(_ops.KERNELS["LANG_SIDE_STREAMS"] == 1 and feats[0].is_cuda
                                                       and cfg.MODEL.DYHEAD.get("LEVEL_STREAMS", True) and not want_raw))
        lang["max_kv"] = max_kv
        trace = [] if want_raw else None                          # raw mode: per-layer tensors, single-stream schedule
        head = pipeline.vldyhead(P, cfg, feats, lang, trace=trace)
        sizes = tuple(tuple(f.shape[-2:]) for f in feats)
        if sizes not in self._anchor_cache:                       # constant per feature-map geometry
            self._anchor_cache[sizes] = pipeline.grid_anchors(P, sizes, cfg.MODEL.RPN.ANCHOR_STRIDE, feats[0].device)
        anchors = self._anchor_cache[sizes]
        post = pipeline.postprocess(cfg, head, anchors, im_wh, tokidx, label_ids, want_cls=want_raw)
        if want_raw:
            return {"post": post, "head": head, "head_trace": trace, "lang": lang, "feats": feats, "anchors": anchors,
                    "vision": vision, "idx": idx, "pooled": pooled}
        if "packed" in post:                                      # fused post-processing hands over the packed block and counts as they are
            packed, counts = post["packed"], post["counts_packed"]
        else:
            packed = torch.cat([post["boxes"], post["scores"][..., None], post["labels"].float()[..., None]], -1)
            # one int per image for the single device->host transfer: live slots, bit 16 = more ties with the K-th score than tie slots
            counts = post["counts"] + post["tie_overflow"].to(post["counts"].dtype) * 65536
        gates = lang["vision_query_gates"]
        if gates is not None:                                     # reference: attn_gate.mean().item() per GCP layer (a float)
            gates = torch.stack([g.float().mean() for g in gates])
        return {"packed": packed, "counts": counts, "gates": gates}

    def _pixels(self, src, dtype):
        """The pixel tensor the device program takes: the caller's fp32 NCHW tensor AS IT IS when the patch-embedding kernel can read it
        (mq_patch_embed_fwd rounds to the operand type itself: no cast pass, no channels-last copy), else 16-bit channels-last pixels."""
        P = self._plan
        if _ops.KERNELS["PATCH_EMBED_FUSED"] == 1 and P.get("_r32") and src.dtype == torch.float32 and src.is_contiguous() \
                and src.shape[1] == 3 and "backbone.body.patch_embed.wpk_nchw" in P:
            return src
        return src.to(dtype).contiguous(memory_format=torch.channels_last)

    def _split_counts(self, counts):
        """Packed per-image counts -> live slots; remembers (and warns about) images whose ties with the K-th score did not fit."""
        self.last_tie_overflow = [bool(c >> 16) for c in counts]
        if any(self.last_tie_overflow):
            import warnings
            warnings.warn(f"{sum(self.last_tie_overflow)} image(s) had more detections tied with the DETECTIONS_PER_IMG-th score than "
                          "MODEL.ATSS.TIE_SLOTS output slots: the reference would return all of them (rpn/inference.py:757-766)")
        return [c & 0xFFFF for c in counts]

    def _full_program(self, x, input_ids, attention_mask, vision, idx, tokidx, label_ids, im_wh, max_kv=0, want_raw=False):
        """Whole device forward.  The image-independent part of the language backbone (embeddings + BERT layers below
        the first GCP block) runs on a side stream under the Swin backbone: its launches are tiny (B x 256 tokens)."""
        P, cfg = self._plan, self.cfg
        front = None
        if x.is_cuda and cfg.MODEL.DYHEAD.get("LEVEL_STREAMS", True) and _ops.KERNELS["FRONT_SIDE_STREAM"] == 1:
            main, text = torch.cuda.current_stream(), pipeline._side_streams(x.device, 1, "text")[0]
            text.wait_stream(main)
            with torch.cuda.stream(text):
                front = pipeline.language_front(P, cfg, input_ids, attention_mask, vision is not None, max_kv=max_kv)
        feats, pooled = self._backbone_stage(x)
        if front is not None:
            main.wait_stream(text)
        else:
            front = pipeline.language_front(P, cfg, input_ids, attention_mask, vision is not None, max_kv=max_kv)
        out = self._head_stage(feats, pooled, front, input_ids, attention_mask, vision, idx, tokidx, label_ids, im_wh, max_kv,
                               want_raw=want_raw)
        if not want_raw:
            out.update(feats=feats, pooled=pooled, front=front)
        return out

    def _staggered_program(self, x, input_ids, attention_mask, vision, idx, tokidx, label_ids, im_wh, max_kv=0):
        """The whole device forward with the batch cut into MODEL.MICRO_BATCHES lanes whose stages are issued as a SOFTWARE PIPELINE.
        Why: a forward is two runs of chip-filling kernels (Swin + FPN; the VLDyHead layers) separated / followed by long CHAINS of
        small launches that cannot use the chip -- the image-dependent half of the language backbone (pre-select, 6 GCP blocks + 6 BERT
        layers on B x 256 tokens: ~170 launches between FPN and the first fusion layer, ~2 ms at B = 8) and the post-processing (~145
        launches after the last one, ~1 ms).  With lanes a stage apart the chains of one lane run under the big kernels of another:

            main stream :  Swin/FPN(0)  Swin/FPN(1)  ...  head(0)        head(1)        ...
            stream L_m  :               language rest(0) | language rest(1)  (waits for FPN(m), feeds head(m))
            stream P_m  :                                                post(0) under head(1) ...   (the last lane's runs exposed)

        Every side stream forks from the CAPTURING stream (a fork from a forked stream crashes hipStreamEndCapture on ROCm 7.0 --
        tools/graph_probe.py, GPU call 4 of round 4), lanes only meet through events between streams that are already in the capture.
        Per-image results do not depend on the lane split: no kernel mixes batch items."""
        P, cfg = self._plan, self.cfg
        M = self.micro_batches
        Bn = x.shape[0]
        per = -(-Bn // M)
        cuts = [(a, min(a + per, Bn)) for a in range(0, Bn, per)]
        n = len(cuts)
        cut = lambda t, a, b, batched=True: t[a:b] if (t is not None and batched) else t      # noqa: E731
        dev = x.device
        main = torch.cuda.current_stream()
        S = pipeline._side_streams
        Fs, Ls, Ps = S(dev, n, "lane_front"), S(dev, n, "lane_lang"), S(dev, n, "lane_post")
        hold = []                                            # tensors that cross streams stay alive until every stream has joined `main`
        args = [dict(ids=cut(input_ids, a, b), am=cut(attention_mask, a, b), vision=cut(vision, a, b), idx=cut(idx, a, b),
                     tokidx=cut(tokidx, a, b, tokidx.dim() == 3), labels=cut(label_ids, a, b, label_ids.dim() == 2), wh=im_wh[a:b])
                for (a, b) in cuts]
        fronts, langs, outs = [None] * n, [None] * n, [None] * n
        for m in range(n):                                   # image-independent BERT layers: tiny launches beside the first Swin
            Fs[m].wait_stream(main)
            with torch.cuda.stream(Fs[m]):
                fronts[m] = pipeline.language_front(P, cfg, args[m]["ids"], args[m]["am"], vision is not None, max_kv=max_kv)
        feats = [None] * n
        for m, (a, b) in enumerate(cuts):
            f, pooled = self._backbone_stage(x[a:b])
            feats[m] = f
            Ls[m].wait_stream(main)                          # FPN(m) done
            Ls[m].wait_stream(Fs[m])
            with torch.cuda.stream(Ls[m]):
                langs[m] = pipeline.language_backbone(P, cfg, args[m]["ids"], args[m]["am"], args[m]["vision"], pooled, args[m]["idx"],
                                                      want_gates=False, front=fronts[m], max_kv=max_kv)
                langs[m]["max_kv"] = max_kv
            hold.append((f, pooled, fronts[m], langs[m]))
        sizes = tuple(tuple(t.shape[-2:]) for t in feats[0])
        if sizes not in self._anchor_cache:
            self._anchor_cache[sizes] = pipeline.grid_anchors(P, sizes, cfg.MODEL.RPN.ANCHOR_STRIDE, dev)
        anchors = self._anchor_cache[sizes]
        for m in range(n):
            main.wait_stream(Ls[m])
            head = pipeline.vldyhead(P, cfg, feats[m], langs[m])
            hold.append(head)
            last = m == n - 1
            st = main if last else Ps[m]
            if not last:
                st.wait_stream(main)
            with torch.cuda.stream(st):                      # on a side stream: levels one after the other (no fork from a forked stream)
                post = pipeline.postprocess(cfg, head, anchors, args[m]["wh"], args[m]["tokidx"], args[m]["labels"], level_streams=last)
                if "packed" in post:
                    outs[m] = (post["packed"], post["counts_packed"])
                else:
                    packed = torch.cat([post["boxes"], post["scores"][..., None], post["labels"].float()[..., None]], -1)
                    outs[m] = (packed, post["counts"] + post["tie_overflow"].to(post["counts"].dtype) * 65536)
            hold.append(post)
        for m in range(n - 1):
            main.wait_stream(Ps[m])
        for m in range(n):
            main.wait_stream(Fs[m])
        res = {"packed": torch.cat([o[0] for o in outs]), "counts": torch.cat([o[1] for o in outs]), "gates": None, "_hold": hold}
        return res

    def _rest_program(self, feats, pooled, front, input_ids, attention_mask, vision, idx, tokidx, label_ids, im_wh, max_kv=0):
        """Device forward from cached Swin / FPN features and a cached language front (the LVIS protocol
        sends the same pixels 31 times, engine/inference.py:605-625, and the same 31 captions for every image)."""
        out = self._head_stage(list(feats), pooled, front, input_ids, attention_mask, vision, idx, tokidx, label_ids, im_wh, max_kv)
        out.update(feats=list(feats), pooled=pooled, front=front)
        return out

    def clear_caches(self):
        """Drop the per-image feature cache, the per-caption language cache and every captured graph."""
        self._feat_cache = None
        self._front_cache = OrderedDict()
        self._drop_graphs()

    @torch.no_grad()
    def forward(self, images, targets=None, captions=None, positive_map=None, greenlight_map=None,
                return_backbone_features=False, return_raw=False, input_ids=None, attention_mask=None, reuse_backbone=None):
        """generalized_vl_rcnn_new.py:307-519, eval.  `reuse_backbone` (not in the reference; keyword only in spirit): None = follow
        MODEL.BACKBONE_CACHE; False = this call recomputes Swin / FPN whatever the cache holds.  The cache (f1) recognises the
        previous call's pixels by OBJECT IDENTITY + torch's version counter of `images.tensors`: a producer that refills the same
        tensor WITHOUT bumping the counter (numpy / DLPack views of its memory, `.data` writes, IPC or custom-kernel writers) must
        pass reuse_backbone=False (or call `clear_caches()` / set MODEL.BACKBONE_CACHE False) -- nothing else can see such a write."""
        if self.training:
            raise NotImplementedError("training forward is out of scope")
        images = to_image_list(images)
        dev = images.tensors.device
        if self._plan is None or self._plan_key != dev:
            self.prepare(dev)
        _ops.activate(self._kernels)
        P, cfg = self._plan, self.cfg
        dtype = P["backbone.body.patch_embed.proj.weight"].dtype
        Bn = images.tensors.shape[0]
        if input_ids is None:
            input_ids, attention_mask, max_kv = self.tokenize(captions, dev)
            cap_key = tuple(captions)
        else:                                                     # caller-supplied ids (tests): one host sync
            max_kv = int((attention_mask.cpu() * torch.arange(1, attention_mask.shape[1] + 1)).max())
            cap_key = None
        T = input_ids.shape[1]

        # host-side glue: all memoised, no device sync.  Token positions that the tokenizer's truncation to MAX_QUERY_LEN cut
        # away cannot be scored (the reference indexes a [L, T] map with them, generalized_vl_rcnn_new.py:295-305): dropped
        if any(t >= T for v in positive_map.values() for t in (v if not isinstance(v, int) else [v])):
            positive_map = {k: [t for t in (v if not isinstance(v, int) else [v]) if t < T] for k, v in positive_map.items()}
        labels_in_caption = [k for k, v in positive_map.items() if len(v) != 0]
        pm_key = tuple((k, tuple(positive_map[k])) for k in labels_in_caption)
        onehot = str(cfg.MODEL.DYHEAD.get("SCORE_AGG", "MEAN")).upper() == "ONEHOT"
        if max_kv > 0:
            # the alignment kernel scores text columns below 16 ceil(max_kv / 16) only: a positive_map that names a token behind the
            # caption's last live one (abnormal, but legal for the reference, which scores all T columns) widens the bound instead of
            # being scored as 0 inside a MEAN
            max_kv = max(max_kv, len(positive_map) if onehot else 1 + max((t for k in labels_in_caption for t in positive_map[k]), default=-1))
        # LIVE-ROW COMPACTION (round 5): everything behind the tokenizer runs on the first Tl = 16 ceil(max_kv / 16) text positions instead of
        # the MAX_QUERY_LEN = 256 the caption is padded to.  Padded positions are masked KEYS everywhere (BERT, GCP, VLFuse) and are never
        # scored, so no live output depends on them -- but as ROWS they went through every text-side GEMM, LayerNorm and elementwise kernel
        # (141-token caption: 44 % of those rows).  The key-length bucket is part of the HIP-graph key already, so Tl adds no graph.
        Tl = self._live_len(T, max_kv)
        if Tl < T:
            input_ids, attention_mask = self._live_slice(input_ids, attention_mask, Tl, cap_key, dev)
        vision = idx = None
        if self._use_vq():
            vision, idx = self.query_selector.select_cached(pm_key, labels_in_caption, positive_map, Bn, Tl, dev, dtype)
            if vision.shape[1] == 0:                              # no label of this caption has a vision query: text only
                vision = idx = None
        tk = (len(positive_map), "onehot", str(dev)) if onehot else (pm_key, str(dev))
        hit = self._tokidx_cache.get(tk)
        if hit is None:
            if len(self._tokidx_cache) > 256:
                self._tokidx_cache.clear()
            if onehot:      # scores = logits[:, :, :len(positive_map)] (rpn/inference.py:789-791): class column j is token j, label j + 1
                n = len(positive_map)
                hit = build_token_index({j + 1: [j] for j in range(n)}, list(range(1, n + 1)), dev)
            else:
                hit = build_token_index(positive_map, labels_in_caption, dev)
            self._tokidx_cache[tk] = memoised(hit)
        tokidx, label_ids = hit
        wh_key = (tuple(images.image_sizes), str(dev))
        im_wh = self._wh_cache.get(wh_key)
        if im_wh is None:
            if len(self._wh_cache) > 256:
                self._wh_cache.clear()
            im_wh = self._wh_cache[wh_key] = memoised(torch.tensor([[w, h] for (h, w) in images.image_sizes],
                                                                   dtype=torch.float32, device=dev))
        tail = (input_ids, attention_mask, vision, idx, tokidx, label_ids, im_wh, max_kv)
        if return_raw:
            x = images.tensors.to(dtype).contiguous(memory_format=torch.channels_last)
            raw = self._full_program(x, *tail, want_raw=True)
            return self._pad_raw_text(raw, T) if Tl < T else raw
        from .. import ops
        use_graph = self.use_hip_graph and not ops.timing_active()

        # f1: the pixels of the previous call (same tensor object, not modified since) -> cached Swin / FPN features;
        # a caption seen before -> cached image-independent BERT layers.  The strong reference to the input tensor keeps
        # its storage alive, so object identity + version counter cannot alias a different batch.
        fc = self._feat_cache if (self.backbone_cache and reuse_backbone is not False) else None
        src = images.tensors
        if fc is not None and fc["src"] is src and fc["version"] == src._version and (vision is None or fc["pooled"] is not None):
            self.cache_stats["backbone_hit"] += 1
            fkey = (cap_key, Bn, vision is not None, Tl)
            front = self._front_cache.get(fkey) if cap_key is not None else None
            if front is None:
                self.cache_stats["front_miss"] += 1
                front = pipeline.language_front(P, cfg, input_ids, attention_mask, vision is not None, max_kv=max_kv)
                if cap_key is not None:
                    self._front_cache[fkey] = front
                    while len(self._front_cache) > int(cfg.MODEL.get("LANG_FRONT_CACHE", 64)):
                        self._front_cache.popitem(last=False)
            else:
                self.cache_stats["front_hit"] += 1
                self._front_cache.move_to_end(fkey)
            out = self._run("_rest_program", (fc["feats"], fc["pooled"], front) + tail, use_graph)
        else:
            x = self._pixels(src, dtype)
            staggered = (self.micro_batches > 1 and Bn > 1 and not self.backbone_cache and x.is_cuda
                         and not cfg.VISION_QUERY.RETURN_ATTN_GATE_VALUE and not return_backbone_features
                         and cfg.MODEL.DYHEAD.get("LEVEL_STREAMS", True))
            out = self._run("_staggered_program" if staggered else "_full_program", (x,) + tail, use_graph)
            if self.backbone_cache:
                self.cache_stats["backbone_miss"] += 1
                keep = self._tree_map(lambda t: t.clone(), {"feats": out["feats"], "pooled": out["pooled"], "front": out["front"]})
                self._feat_cache = {"src": src, "version": src._version, "feats": keep["feats"], "pooled": keep["pooled"]}
                if cap_key is not None:
                    self._front_cache[(cap_key, Bn, vision is not None, Tl)] = keep["front"]

        # fixed-shape detections [B, K, 6] for the RCCL all-gather (mq_det_amd.parallel.gather_detections); cloned: under
        # HIP-graph replay `out` are the graph's static buffers, which the next forward overwrites
        self.last_packed = packed = out["packed"].clone()
        gates = out["gates"]
        if gates is not None:
            counts, gates = torch.cat([out["counts"].float(), gates.float()]).tolist(), None
            counts, gates = [int(c) for c in counts[:Bn]], counts[Bn:]
        else:
            counts = out["counts"].tolist()                       # the one device->host sync of the forward
        counts = self._split_counts(counts)
        result = self._boxlists(packed, counts, images.image_sizes)
        if cfg.VISION_QUERY.RETURN_ATTN_GATE_VALUE:
            return result, gates
        if return_backbone_features:
            return result, [f.float().contiguous() for f in out["feats"]]
        return result

    @staticmethod
    def _boxlists(packed, counts, image_sizes, first=0):
        """[items, K, 6] packed detections (a tensor nobody else writes) + live counts -> list[BoxList]: three launches for the whole batch (boxes,
        scores and int64 labels as contiguous tensors), the BoxLists hold per-image VIEWS of them (round 4: three launches per image)."""
        boxes, scores, labels = packed[..., :4].contiguous(), packed[..., 4].contiguous(), packed[..., 5].to(torch.int64)
        result = []
        for b, (h, w) in enumerate(image_sizes):
            it, n = first + b, counts[first + b]
            bl = BoxList(boxes[it, :n], (int(w), int(h)), mode="xyxy")
            bl.add_field("labels", labels[it, :n])
            bl.add_field("scores", scores[it, :n])
            result.append(bl)
        return result

    # ------------------------------------------------------------------ chunk batching
    def _features(self, images):
        """(feats, pooled) of an image batch through the per-image cache (Swin + FPN run once per distinct tensor)."""
        dev = images.tensors.device
        if self._plan is None or self._plan_key != dev:
            self.prepare(dev)
        _ops.activate(self._kernels)
        src = images.tensors
        fc = self._feat_cache if self.backbone_cache else None
        if fc is not None and fc["src"] is src and fc["version"] == src._version and (fc["pooled"] is not None or not self._use_vq()):
            self.cache_stats["backbone_hit"] += 1
            return fc["feats"], fc["pooled"]
        self.cache_stats["backbone_miss"] += 1
        dtype = self._plan["backbone.body.patch_embed.proj.weight"].dtype
        x = self._pixels(src, dtype)
        feats, pooled = self._backbone_stage(x)
        if self.backbone_cache:
            self._feat_cache = {"src": src, "version": src._version, "feats": feats, "pooled": pooled}
        return feats, pooled

    def _front_for(self, caption, dev, use_vq):
        """Image-independent BERT layers of ONE caption (batch 1), cached per caption."""
        key = ((caption,), 1, use_vq)
        fr = self._front_cache.get(key)
        if fr is None:
            self.cache_stats["front_miss"] += 1
            ids, am, kv = self.tokenize([caption], dev)
            fr = self._front_cache[key] = pipeline.language_front(self._plan, self.cfg, ids, am, use_vq, max_kv=kv)
            while len(self._front_cache) > int(self.cfg.MODEL.get("LANG_FRONT_CACHE", 64)):
                self._front_cache.popitem(last=False)
        else:
            self.cache_stats["front_hit"] += 1
            self._front_cache.move_to_end(key)
        return fr

    @torch.no_grad()
    def forward_chunks(self, images, chunks, max_items=32):
        """The LVIS / ODinW evaluation protocol in ONE call: `chunks` = [(caption, positive_map), ...] (the 31 chunk captions of
        engine/inference.py:605-625) for the SAME image batch.  Equivalent to `[model(images, captions=[c] * B,
        positive_map=pm) for c, pm in chunks]`, but Swin + FPN run once, the image-independent BERT layers once per caption
        (cached across images), and the image-dependent rest runs with the chunks stacked along the batch dimension
        (up to `max_items` image x chunk items per launch sequence): B = 1 -- the reference's TEST.IMS_PER_BATCH -- no longer
        means a batch-1 forward.  Returns list (over chunks) of list[BoxList] (over images)."""
        if self.training:
            raise NotImplementedError("training forward is out of scope")
        images = to_image_list(images)
        dev = images.tensors.device
        feats, pooled = self._features(images)
        P, cfg = self._plan, self.cfg
        dtype = P["backbone.body.patch_embed.proj.weight"].dtype
        Bn = images.tensors.shape[0]
        use_vq = self._use_vq()
        from .. import ops
        use_graph = self.use_hip_graph and not ops.timing_active()
        im_wh = torch.tensor([[w, h] for (h, w) in images.image_sizes], dtype=torch.float32, device=dev)
        results = []
        per = max(1, int(max_items) // Bn)
        for g0 in range(0, len(chunks), per):
            grp = chunks[g0:g0 + per]
            g = len(grp)
            ids, ams, kvs, fronts, pms, labs = [], [], [], [], [], []
            for cap, pm in grp:
                i, a, kv = self.tokenize([cap], dev)
                T = i.shape[1]
                if any(t >= T for v in pm.values() for t in (v if not isinstance(v, int) else [v])):
                    pm = {k: [t for t in (v if not isinstance(v, int) else [v]) if t < T] for k, v in pm.items()}
                ids.append(i)
                ams.append(a)
                kv_pm = max(kv, 1 + max((t for v in pm.values() for t in (v if not isinstance(v, int) else [v])), default=-1))
                if str(cfg.MODEL.DYHEAD.get("SCORE_AGG", "MEAN")).upper() == "ONEHOT":
                    kv_pm = max(kv_pm, len(pm))                   # class column j scores token j (forward() widens the same way)
                kvs.append(min(kv_pm, T) if kv > 0 else kv)
                pms.append(pm)
                labs.append([k for k, v in pm.items() if len(v) != 0])
            T = self._live_len(ids[0].shape[1], max(kvs) if min(kvs) > 0 else 0)          # live-row compaction (see forward): the group's longest caption
            live = lambda t_: t_[:, :T]                           # noqa: E731
            ids, ams = [live(i) for i in ids], [live(a) for a in ams]
            vision = idx = None
            if use_vq:                                            # item order: chunk-major (item = chunk * B + image)
                vision, idx = self.query_selector.select([l for l in labs for _ in range(Bn)], [pm for pm in pms for _ in range(Bn)],
                                                         T, dev, dtype)
                if vision.shape[1] == 0:
                    vision = idx = None
            for cap, _ in grp:
                fronts.append(self._front_for(cap, dev, vision is not None))
            rep = lambda t: t.repeat_interleave(Bn, 0)            # noqa: E731  [g, ...] -> [g * B, ...] chunk-major
            # (the cached fronts hold all MAX_QUERY_LEN positions of their caption: cut to the group's live length here)
            front = {"x": rep(torch.cat([live(f["x"]) for f in fronts])),
                     "x32": None if fronts[0].get("x32") is None else rep(torch.cat([live(f["x32"]) for f in fronts])),
                     "hidden": [rep(torch.cat([live(f["hidden"][k]) for f in fronts])) for k in range(len(fronts[0]["hidden"]))],
                     "key_bias": rep(torch.cat([live(f["key_bias"]) for f in fronts])), "kv_len": rep(torch.cat([f["kv_len"] for f in fronts])),
                     "next": fronts[0]["next"]}
            if str(cfg.MODEL.DYHEAD.get("SCORE_AGG", "MEAN")).upper() == "ONEHOT":      # class column j = token j, label j + 1
                smaps = [({j + 1: [j] for j in range(min(len(pm), T))}, list(range(1, min(len(pm), T) + 1))) for pm in pms]
            else:
                smaps = list(zip(pms, labs))
            L = max(1, max(len(l) for _, l in smaps))
            MT = max(1, max((len(pm[k]) for pm, l in smaps for k in l), default=1))
            tok3 = torch.full((g, L, MT), -1, dtype=torch.int32)
            lab2 = torch.zeros(g, L, dtype=torch.int32)
            for c, (pm, l) in enumerate(smaps):
                for j, k in enumerate(l):
                    tok3[c, j, :len(pm[k])] = torch.tensor(pm[k], dtype=torch.int32)
                    lab2[c, j] = k
            tok3, lab2 = rep(tok3.to(dev)).contiguous(), rep(lab2.to(dev)).contiguous()
            tile = lambda f: f.permute(0, 2, 3, 1).repeat(g, 1, 1, 1).permute(0, 3, 1, 2)     # noqa: E731  NHWC memory kept
            inputs = ([tile(f) for f in feats], None if pooled is None else pooled.repeat(g, 1, 1), front, rep(torch.cat(ids)),
                      rep(torch.cat(ams)), vision, idx, tok3, lab2, im_wh.repeat(g, 1), max(kvs))
            out = self._run("_rest_program", inputs, use_graph)
            packed = out["packed"].clone()
            counts = self._split_counts(out["counts"].tolist())
            for c in range(g):
                results.append(self._boxlists(packed, counts, images.image_sizes, first=c * Bn))
        return results


_DETECTION_META_ARCHITECTURES = {"GeneralizedVLRCNN_New": GeneralizedVLRCNN_New}


def build_detection_model(cfg, **kwargs):
    """modeling/detector/__init__.py:9-14."""
    if cfg.get("GROUNDINGDINO", {}).get("enabled", False):       # groundingdino_new.models.build_model(cfg.GROUNDINGDINO, cfg)
        from .gdino import GroundingDINO
        return GroundingDINO(cfg, **kwargs)
    return _DETECTION_META_ARCHITECTURES[cfg.MODEL.META_ARCHITECTURE](cfg, **kwargs)
