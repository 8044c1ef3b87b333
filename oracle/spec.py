"""Model hyper-parameters used by the oracle (test infrastructure, see oracle/__init__.py).

Values follow the reference configs:
  configs/pretrain/mq-glip-t.yaml, configs/vision_query_5shot/lvis_minival.yaml and
  maskrcnn_benchmark/config/defaults.py (SWINT :720-731, DYHEAD :440-532, ATSS :407-436,
  LANGUAGE_BACKBONE :264-287, VISION_QUERY :899-938, RPN anchors :553-557).
"""
from dataclasses import dataclass, replace
from typing import Tuple


@dataclass(frozen=True)
class Spec:
    # Swin (defaults.py:720-731)
    swin_embed: int = 96
    swin_depths: Tuple[int, ...] = (2, 2, 6, 2)
    swin_heads: Tuple[int, ...] = (3, 6, 12, 24)
    window: int = 7
    mlp_ratio: int = 4
    # FPN
    fpn_out: int = 256
    # BERT-base (HF BertConfig defaults)
    vocab: int = 30522
    max_pos: int = 512
    bert_layers: int = 12
    bert_hidden: int = 768
    bert_heads: int = 12
    bert_inter: int = 3072
    bert_eps: float = 1e-12
    max_query_len: int = 256
    n_lang_layers: int = 1          # MODEL.LANGUAGE_BACKBONE.N_LAYERS
    # GCP (modeling_bert_new.py:522-543, 642-659)
    vision_query: bool = True
    qv_start: int = 6
    gcp_heads: int = 8
    gcp_dim_head: int = 64
    pre_dim_head: int = 32
    pre_layers: int = 2
    ff_mult: int = 4
    vision_scale: float = 1.0
    num_query_per_class: int = 5
    # VLDyHead
    dyhead_convs: int = 6
    dyhead_channels: int = 256
    fuse_embed: int = 2048
    fuse_heads: int = 8
    gn_groups: int = 16
    gn_eps: float = 1e-5
    num_classes: int = 1204         # MODEL.DYHEAD.NUM_CLASSES (cls_logits width = num_classes-1)
    prior_prob: float = 0.01
    log_scale: float = 0.0
    # anchors / post-process
    anchor_sizes: Tuple[int, ...] = (64, 128, 256, 512, 1024)
    anchor_strides: Tuple[int, ...] = (8, 16, 32, 64, 128)
    pre_nms_thresh: float = 0.05
    pre_nms_top_n: int = 1000
    nms_thresh: float = 0.6
    detections_per_img: int = 300
    mdetr_class_num: int = 3000     # TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM (-1 => use num_classes-1)
    score_agg: str = "MEAN"         # MODEL.DYHEAD.SCORE_AGG: MEAN | MAX | ONEHOT | POWER (MDETR-style aggregation only)
    size_divisibility: int = 32

    @property
    def swin_dims(self):
        return tuple(self.swin_embed * 2 ** i for i in range(len(self.swin_depths)))


def glip_t_spec(**kw) -> Spec:
    """MQ-GLIP-T (BASELINE.json configs[1])."""
    return replace(Spec(), **kw)


def tiny_spec(**kw) -> Spec:
    """Same widths / head dims as MQ-GLIP-T (kernels see the real tile shapes) but shallow,
    with a small vocabulary, so the full model runs in seconds on CPU."""
    base = Spec(swin_depths=(2, 2, 2, 2), bert_layers=4, qv_start=2, dyhead_convs=2,
                vocab=2048, num_classes=81, mdetr_class_num=-1)
    return replace(base, **kw)


def glip_l_spec(**kw) -> Spec:
    """MQ-GLIP-L (BASELINE.json configs[3]; configs/pretrain/mq-glip-l.yaml:11-17,41): Swin-L (embed 192, depths 2-2-18-2,
    heads 6-12-24-48, window 12), 8 fusion layers; the language model stays bert-base (:21-22)."""
    return replace(Spec(swin_embed=192, swin_depths=(2, 2, 18, 2), swin_heads=(6, 12, 24, 48), window=12, dyhead_convs=8), **kw)


def tiny_l_spec(**kw) -> Spec:
    """Swin-L widths / heads / window 12 and the real head dims, shallow (depths 2-2-2-2, 2 fusion layers)."""
    base = Spec(swin_embed=192, swin_depths=(2, 2, 2, 2), swin_heads=(6, 12, 24, 48), window=12, bert_layers=4, qv_start=2,
                dyhead_convs=2, vocab=2048, num_classes=81, mdetr_class_num=-1)
    return replace(base, **kw)


@dataclass(frozen=True)
class GdinoSpec(Spec):
    """MQ-GroundingDINO-T (BASELINE.json configs[4]): defaults.py:944-1001 + configs/pretrain/mq-groundingdino-t.yaml."""
    hidden: int = 256               # GROUNDINGDINO.hidden_dim
    nheads: int = 8
    enc_layers: int = 6
    dec_layers: int = 6
    ffn: int = 2048                 # dim_feedforward; fusion embed / text-enhancer FFN = ffn // 2, their heads = nheads // 2
    num_queries: int = 900
    levels: int = 4                 # num_feature_levels (Swin stages 1-3 + one stride-2 conv)
    points: int = 4                 # enc_n_points == dec_n_points
    pe_temperature: float = 20.0    # pe_temperatureH == pe_temperatureW
    box_threshold: float = 0.05
    max_text_len: int = 256
    fusion_init: float = 1e-4       # BiAttentionBlock init_values (layer scale)
    gn_groups: int = 32             # input_proj GroupNorm(32, hidden)
    num_classes: int = 81           # MODEL.DYHEAD.NUM_CLASSES (defaults.py): width of the class-score tensor - 1


def gdino_t_spec(**kw) -> GdinoSpec:
    return replace(GdinoSpec(), **kw)


def tiny_gdino_spec(**kw) -> GdinoSpec:
    """Real widths / head dims, shallow: Swin depths 2-2-2-2, BERT 7 layers with one GCP block (the reference hard-codes
    start_qv_layer_index = 6, modeling_bert_new.py:532), 2 encoder + 2 decoder layers, 40 queries, small vocabulary."""
    base = GdinoSpec(swin_depths=(2, 2, 2, 2), bert_layers=7, qv_start=6, vocab=2048, enc_layers=2, dec_layers=2, num_queries=40)
    return replace(base, **kw)
