"""Minimal yacs-compatible config node + the subset of the reference's config tree that the
inference hot path reads.

In the reference environment the real yacs `cfg` (maskrcnn_benchmark/config/defaults.py) can be passed
to `build_detection_model` unchanged -- the modules here only do attribute reads.  yacs is not installed
in the build / GPU image, hence this ~100-line stand-in (attribute access, YAML / list merging, freeze).
Key names and default values follow defaults.py (line numbers in comments) + configs/pretrain/mq-glip-t.yaml.
"""
import ast
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init=None, new_allowed=True):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"cfg is frozen, cannot set {k}")
        self[k] = v

    def freeze(self):
        object.__setattr__(self, "_frozen", True)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        object.__setattr__(self, "_frozen", False)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        return out

    @staticmethod
    def _coerce(v):
        if isinstance(v, str) and v[:1] in "([":
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k].merge_from_other_cfg(v)
            else:
                dict.__setitem__(self, k, CfgNode(v) if isinstance(v, dict) else self._coerce(v))

    def merge_from_file(self, path):
        with open(path) as f:
            self.merge_from_other_cfg(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, CfgNode())
            if isinstance(val, str):
                try:
                    val = ast.literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            dict.__setitem__(node, parts[-1], val)


def get_cfg():
    """Defaults for the MQ-GLIP-T inference path (defaults.py + configs/pretrain/mq-glip-t.yaml +
    configs/vision_query_5shot/lvis_minival.yaml where noted)."""
    C = CfgNode
    cfg = C()
    cfg.MODEL = C(dict(
        META_ARCHITECTURE="GeneralizedVLRCNN_New", DEVICE="cuda", RPN_ONLY=True, RPN_ARCHITECTURE="VLDYHEAD",
        DEBUG=False, LINEAR_PROB=False, WEIGHT="",
        USE_HIP_GRAPH=True,       # new: replay the device forward as one HIP graph per input-shape signature
        COMPUTE_DTYPE="float16",   # new: "float16" (configs[1]) or "bfloat16" (configs[3]): operand type of the kernels, fp32 accumulate
        BACKBONE=dict(CONV_BODY="SWINT-FPN-RETINANET", OUT_CHANNELS=256, FREEZE_CONV_BODY_AT=-1, FREEZE=False,
                      USE_CHECKPOINT=False, OUT_FEATURES=("stage2", "stage3", "stage4", "stage5")),
        SWINT=dict(EMBED_DIM=96, OUT_CHANNELS=(96, 192, 384, 768), DEPTHS=(2, 2, 6, 2), NUM_HEADS=(3, 6, 12, 24),
                   WINDOW_SIZE=7, MLP_RATIO=4, DROP_PATH_RATE=0.2, APE=False, VERSION="v1"),       # :720-731
        FPN=dict(FREEZE=False, USE_GN=False, USE_RELU=False, DROP_BLOCK=True, USE_SPP=False, USE_PAN=False,
                 USE_DYHEAD=False, RETURN_SWINT_FEATURE_BEFORE_FUSION=False),                        # :291-303
        GROUP_NORM=dict(DIM_PER_GP=-1, NUM_GROUPS=16, EPSILON=1e-5),                                   # :313-319
        LANGUAGE_BACKBONE=dict(FREEZE=False, TOKENIZER_TYPE="bert-base-uncased", MODEL_TYPE="bert-base-uncased",
                               LANG_DIM=768, MAX_QUERY_LEN=256, N_LAYERS=1, PAD_MAX=True, MASK_SPECIAL=False,
                               USE_CHECKPOINT=False, VOCAB_SIZE=0, BERT_VOCAB_SIZE=0, NUM_HIDDEN_LAYERS=12,         # :264-287
                               COMPACT_TEXT=True),      # new: the device programs run on the first 16 ceil(live tokens / 16) text positions only
        RPN=dict(USE_FPN=True, ANCHOR_SIZES=(64, 128, 256, 512, 1024), ANCHOR_STRIDE=(8, 16, 32, 64, 128),
                 ASPECT_RATIOS=(1.0,), SCALES_PER_OCTAVE=1, OCTAVE=2.0, STRADDLE_THRESH=0, FREEZE=False,
                 FORCE_BOXES=False, RETURN_FUSED_FEATURES=False),
        ATSS=dict(NUM_CLASSES=81, PRIOR_PROB=0.01, INFERENCE_TH=0.05, NMS_TH=0.6, PRE_NMS_TOP_N=1000,
                  DETECTIONS_PER_IMG=100),                                                             # :407-436
        DYHEAD=dict(NUM_CLASSES=81, PRIOR_PROB=0.01, NUM_CONVS=6, CHANNELS=256, USE_GN=True, USE_DYRELU=True,
                    USE_DFCONV=True, USE_DYFUSE=True, FUSED_DCN=True, LEVEL_STREAMS=True, SCORE_AGG="MEAN", LOG_SCALE=0.0, USE_CHECKPOINT=False,
                    FUSE_CONFIG=dict(EARLY_FUSE_ON=True, TYPE="MHA-B", USE_DOT_PRODUCT_TOKEN_LOSS=True,
                                     USE_FUSED_FEATURES_DOT_PRODUCT=True, USE_LAYER_SCALE=True,
                                     CLAMP_MIN_FOR_UNDERFLOW=True, CLAMP_MAX_FOR_OVERFLOW=True,
                                     CLAMP_BERTATTN_MIN_FOR_UNDERFLOW=True, CLAMP_BERTATTN_MAX_FOR_OVERFLOW=True,
                                     CLAMP_DOT_PRODUCT=True, SEPARATE_BIDIRECTIONAL=False, STABLE_SOFTMAX_2D=False,
                                     ADD_LINEAR_LAYER=False, MLM_LOSS=False, USE_TOKEN_LOSS=False,
                                     USE_CONTRASTIVE_ALIGN_LOSS=False)),
        ROI_BOX_HEAD=dict(POOLER_RESOLUTION=7, POOLER_SCALES=(0.125, 0.0625, 0.03125, 0.015625, 0.0078125),
                          POOLER_SAMPLING_RATIO=0),
    ))
    cfg.VISION_QUERY = C(dict(                                                                       # :899-938
        ENABLED=True, QUERY_BANK_PATH="", NUM_QUERY_PER_CLASS=5, VISION_SCALE=1.0, SHARE_KV=False,
        SELECT_FPN_LEVEL=True, PURE_TEXT_RATE=0.0, TEXT_DROPOUT=0.0, CONDITION_GATE=True, NONLINEAR_GATE=True,
        NO_CAT=True, FIX_ATTN_GATE=-1.0, ADD_ADAPT_LAYER=False, QUERY_FUSION=False, DISABLE_SELECTOR=False,
        LEARNABLE_BANK=False, ADD_VISION_LAYER=False, RANDOM_KSHOT=False, MASK_DURING_INFERENCE=False,
        AUGMENT_IMAGE_WITH_QUERY=False, RETURN_ATTN_GATE_VALUE=False, EXPAND_RATIO=1.5, MAX_QUERY_NUMBER=5000,
        SIMILARITY_THRESHOLD=0.85))
    cfg.TEST = C(dict(IMS_PER_BATCH=8, CHUNKED_EVALUATION=-1, MDETR_STYLE_AGGREGATE_CLASS_NUM=-1,
                      USE_MULTISCALE=False, EVAL_TASK="detection"))
    cfg.DATALOADER = C(dict(SIZE_DIVISIBILITY=32, NUM_WORKERS=0))
    cfg.DATASETS = C(dict(SEPARATION_TOKENS=". ", ONE_HOT=False))
    cfg.INPUT = C(dict(PIXEL_MEAN=[103.530, 116.280, 123.675], PIXEL_STD=[57.375, 57.120, 58.395],
                       MIN_SIZE_TEST=800, MAX_SIZE_TEST=1333))
    cfg.GLIPKNOW = C(dict(KNOWLEDGE_FILE="", PARALLEL_LANGUAGE_INPUT=False))
    # config/defaults.py:944-1001 (key names are the reference's: the flag system is part of the boundary)
    cfg.GROUNDINGDINO = C(dict(
        enabled=False, modelname="groundingdino", backbone="swin_T_224_1k", position_embedding="sine", pe_temperatureH=20,
        pe_temperatureW=20, return_interm_indices=[1, 2, 3], backbone_freeze_keywords=None, enc_layers=6, dec_layers=6,
        pre_norm=False, dim_feedforward=2048, hidden_dim=256, dropout=0.0, nheads=8, num_queries=900, query_dim=4,
        num_patterns=0, num_feature_levels=4, enc_n_points=4, dec_n_points=4, two_stage_type="standard",
        two_stage_bbox_embed_share=False, two_stage_class_embed_share=False, transformer_activation="relu",
        dec_pred_bbox_embed_share=True, embed_init_tgt=True, max_text_len=256, text_encoder_type="bert-base-uncased",
        use_text_enhancer=True, use_fusion_layer=True, use_checkpoint=False, use_transformer_ckpt=False,
        use_text_cross_attention=True, text_dropout=0.0, fusion_dropout=0.0, fusion_droppath=0.1, sub_sentence_present=True,
        box_threshold=0.05))
    return cfg


def get_gdino_cfg():
    """Defaults + configs/pretrain/mq-groundingdino-t.yaml (the keys the inference forward reads)."""
    cfg = get_cfg()
    cfg.GROUNDINGDINO.enabled = True
    cfg.MODEL.ROI_BOX_HEAD.POOLER_SCALES = (0.125, 0.0625, 0.03125, 0.015625)
    cfg.INPUT.FORMAT = "rgb"
    cfg.INPUT.PIXEL_MEAN, cfg.INPUT.PIXEL_STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    return cfg
