"""Parameter tree of MQ-GLIP with the reference's module / parameter names.

Checkpoints reach the model through the reference's `DetectronCheckpointer` -> suffix matching
(`utils/model_serialization.py:20-101`) -> strict `load_state_dict`, so `state_dict()` keys must equal
the reference's.  The tree below is derived from the reference module definitions:
  backbone.body.*      modeling/backbone/swint.py:77-109,162-184,252-256,398-410,494-552
  backbone.fpn.*       modeling/backbone/fpn.py:33-46,141-146 (+ make_layers.py:95-124, no GN)
  language_backbone.body.model.*   HF BertModel (embeddings / encoder.layer.N) +
                       language_backbone/modeling_bert_new.py:150-160,268-289,390-396,542-543,657-659
  rpn.head.*           modeling/rpn/vldyhead.py:166-188,258-262,635-729, utils/fuse_helper.py:184-195,368-384,
                       modeling/rpn/modeling_bert.py:53-55,178-183,243-265, layers/dyrelu.py:62-67,
                       layers/deform_conv.py:349-358
  rpn.anchor_generator.cell_anchors.*   modeling/rpn/anchor_generator.py:51-68
The modules are bare containers: the forward pass is the functional HIP pipeline in pipeline.py, which
reads an fp16 "plan" derived from these parameters.
"""
import math

import numpy as np
import torch
from torch import nn


def rel_pos_index(ws):
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    pos = torch.stack([ys.reshape(-1), xs.reshape(-1)])
    d = pos[:, :, None] - pos[:, None, :] + (ws - 1)
    return d[0] * (2 * ws - 1) + d[1]


def cell_anchor(stride, size):
    """One square anchor per location (ASPECT_RATIOS (1.0,), SCALES_PER_OCTAVE 1): the stride-sized
    reference window centred on (stride-1)/2, scaled to `size` (anchor_generator.py:356-425)."""
    c = 0.5 * (stride - 1)
    side = np.round(np.sqrt(float(stride) * stride)) * (size / stride)
    half = 0.5 * (side - 1)
    return torch.tensor([[c - half, c - half, c + half, c + half]], dtype=torch.float32)


def bert_vocab_size(cfg):
    """Word-embedding rows of the BERT language backbone.  The reference takes them from HF
    `BertConfig.from_pretrained(MODEL.LANGUAGE_BACKBONE.MODEL_TYPE)` (bert_model_new.py:24: bert-base-uncased = 30522), NOT
    from its `LANGUAGE_BACKBONE.VOCAB_SIZE` key (defaults.py:285, the RNN language model's, default 0).  Same here: a local
    MODEL_TYPE directory with a config.json is read like `from_pretrained` would; the product-only override key is
    BERT_VOCAB_SIZE (tests with small vocabularies)."""
    import json
    import os
    LB = cfg.MODEL.LANGUAGE_BACKBONE
    if LB.get("BERT_VOCAB_SIZE", 0):
        return int(LB.BERT_VOCAB_SIZE)
    for key in ("MODEL_TYPE", "TOKENIZER_TYPE"):
        cj = os.path.join(str(LB.get(key, "")), "config.json")
        if os.path.isfile(cj):
            v = json.load(open(cj)).get("vocab_size")
            if v:
                return int(v)
    return 30522


def swin_specs(SW, p):
    """Swin parameters under prefix `p` (maskrcnn_benchmark swint.py and GroundingDINO swin_transformer.py share the names)."""
    ws = SW.WINDOW_SIZE
    dims = [SW.EMBED_DIM * 2 ** i for i in range(len(SW.DEPTHS))]
    yield p + ".patch_embed.proj.weight", (dims[0], 3, 4, 4), "w"
    yield p + ".patch_embed.proj.bias", (dims[0],), "b"
    yield p + ".patch_embed.norm.weight", (dims[0],), "one"
    yield p + ".patch_embed.norm.bias", (dims[0],), "b"
    for i, (depth, heads) in enumerate(zip(SW.DEPTHS, SW.NUM_HEADS)):
        C = dims[i]
        for j in range(depth):
            b = f"{p}.layers.{i}.blocks.{j}"
            for n in ("norm1", "norm2"):
                yield f"{b}.{n}.weight", (C,), "one"
                yield f"{b}.{n}.bias", (C,), "b"
            yield b + ".attn.qkv.weight", (3 * C, C), "w"
            yield b + ".attn.qkv.bias", (3 * C,), "b"
            yield b + ".attn.proj.weight", (C, C), "w"
            yield b + ".attn.proj.bias", (C,), "b"
            yield b + ".attn.relative_position_bias_table", ((2 * ws - 1) ** 2, heads), "table"
            yield b + ".attn.relative_position_index", (ws * ws, ws * ws), "buf:relidx"
            hid = int(C * SW.MLP_RATIO)
            yield b + ".mlp.fc1.weight", (hid, C), "w"
            yield b + ".mlp.fc1.bias", (hid,), "b"
            yield b + ".mlp.fc2.weight", (C, hid), "w"
            yield b + ".mlp.fc2.bias", (C,), "b"
        if i < len(dims) - 1:
            yield f"{p}.layers.{i}.downsample.norm.weight", (4 * C,), "one"
            yield f"{p}.layers.{i}.downsample.norm.bias", (4 * C,), "b"
            yield f"{p}.layers.{i}.downsample.reduction.weight", (2 * C, 4 * C), "w"
        if i > 0:
            yield f"{p}.norm{i}.weight", (C,), "one"
            yield f"{p}.norm{i}.bias", (C,), "b"


def bert_layer_specs(b, H):
    inter = 4 * H
    for n in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
        yield f"{b}.{n}.weight", (H, H), "w"
        yield f"{b}.{n}.bias", (H,), "b"
    yield b + ".attention.output.LayerNorm.weight", (H,), "one"
    yield b + ".attention.output.LayerNorm.bias", (H,), "b"
    yield b + ".intermediate.dense.weight", (inter, H), "w"
    yield b + ".intermediate.dense.bias", (inter,), "b"
    yield b + ".output.dense.weight", (H, inter), "w"
    yield b + ".output.dense.bias", (H,), "b"
    yield b + ".output.LayerNorm.weight", (H,), "one"
    yield b + ".output.LayerNorm.bias", (H,), "b"


def language_specs(cfg, p, O):
    """BERT-base + GCP (QVBertModel) parameters under prefix `p`; O = width of the image tokens pre-select attends to."""
    M = cfg.MODEL
    LB = M.LANGUAGE_BACKBONE
    H = LB.LANG_DIM

    def bert_layer(b):
        yield from bert_layer_specs(b, H)
    yield p + ".embeddings.word_embeddings.weight", (bert_vocab_size(cfg), H), "emb"
    yield p + ".embeddings.position_embeddings.weight", (512, H), "emb"
    yield p + ".embeddings.token_type_embeddings.weight", (2, H), "emb"
    yield p + ".embeddings.LayerNorm.weight", (H,), "one"
    yield p + ".embeddings.LayerNorm.bias", (H,), "b"

    nl = LB.get("NUM_HIDDEN_LAYERS", 12)
    for i in range(nl):
        yield from bert_layer(f"{p}.encoder.layer.{i}")
    if cfg.VISION_QUERY.ENABLED:
        qv_start = LB.get("QV_START", 6)
        inner = 8 * 64
        for i in range(nl - qv_start):
            b = f"{p}.encoder.qv_layer.{i}"
            for n in ("attn.norm", "attn.norm_kv", "attn_gate.norm", "ff.norm"):
                yield f"{b}.{n}.weight", (H,), "one"
                yield f"{b}.{n}.bias", (H,), "b"
            yield b + ".attn.to_q.weight", (inner, H), "w"
            yield b + ".attn.to_kv.weight", (2 * inner, H), "w"
            yield b + ".attn.to_out.weight", (H, inner), "w"
            yield b + ".attn_gate.linear1.weight", (H // 2, H), "w"
            yield b + ".attn_gate.linear2.weight", (1, H // 2), "zero"
            yield b + ".ff.linear1.weight", (4 * H, H), "w"
            yield b + ".ff.linear2.weight", (H, 4 * H), "w"
            yield b + ".ff_gate", (1,), "zero"
        Cv, pin = O, 8 * 32
        for i in range(2):
            b = f"{p}.pre_select.layers.{i}"
            out = Cv if i == 0 else H
            for n, c in (("image_condition.norm", Cv), ("image_condition.norm_kv", Cv), ("ff.norm", out)):
                yield f"{b}.{n}.weight", (c,), "one"
                yield f"{b}.{n}.bias", (c,), "b"
            yield b + ".image_condition.to_q.weight", (pin, Cv), "w"
            yield b + ".image_condition.to_kv.weight", (2 * pin, Cv), "w"
            yield b + ".image_condition.to_out.weight", (out, pin), "w"
            yield b + ".ff.linear1.weight", (4 * out, out), "w"
            yield b + ".ff.linear2.weight", (out, 4 * out), "w"
            if out != Cv:
                yield b + ".res_mapping.weight", (out, Cv), "w"


def param_specs(cfg):
    """Yield (name, shape, kind) for every parameter / buffer.  kind: w (weight, fan-in scaled),
    b (zero), one, zero, table, buf:<tag>."""
    M = cfg.MODEL
    dims = [M.SWINT.EMBED_DIM * 2 ** i for i in range(len(M.SWINT.DEPTHS))]
    yield from swin_specs(M.SWINT, "backbone.body")
    p = "backbone.fpn"
    O = M.BACKBONE.OUT_CHANNELS
    for idx, cin in ((2, dims[-3]), (3, dims[-2]), (4, dims[-1])):
        yield f"{p}.fpn_inner{idx}.weight", (O, cin, 1, 1), "w"
        yield f"{p}.fpn_inner{idx}.bias", (O,), "b"
        yield f"{p}.fpn_layer{idx}.weight", (O, O, 3, 3), "w"
        yield f"{p}.fpn_layer{idx}.bias", (O,), "b"
    for n in ("p6", "p7"):
        yield f"{p}.top_blocks.{n}.weight", (O, O, 3, 3), "w"
        yield f"{p}.top_blocks.{n}.bias", (O,), "b"

    # ---- language backbone
    LB = M.LANGUAGE_BACKBONE
    H = LB.LANG_DIM
    inter = 4 * H
    yield from language_specs(cfg, "language_backbone.body.model", O)

    def bert_layer(b):
        yield from bert_layer_specs(b, H)

    # ---- VLDyHead
    p = "rpn.head"
    D = M.DYHEAD
    C, E = D.CHANNELS, 2048
    for i in range(D.NUM_CONVS):
        b = f"{p}.dyhead_tower.{3 * i}.b_attn"
        yield b + ".layer_norm_v.weight", (C,), "one"
        yield b + ".layer_norm_v.bias", (C,), "b"
        yield b + ".layer_norm_l.weight", (H,), "one"
        yield b + ".layer_norm_l.bias", (H,), "b"
        for n, (o, c) in (("v_proj", (E, C)), ("l_proj", (E, H)), ("values_v_proj", (E, C)), ("values_l_proj", (E, H)),
                          ("out_v_proj", (C, E)), ("out_l_proj", (H, E))):
            yield f"{b}.attn.{n}.weight", (o, c), "w"
            yield f"{b}.attn.{n}.bias", (o,), "b"
        yield b + ".gamma_v", (C,), f"const:{1.0 / D.NUM_CONVS}"
        yield b + ".gamma_l", (H,), f"const:{1.0 / D.NUM_CONVS}"
        yield from bert_layer(f"{p}.dyhead_tower.{3 * i + 1}")
        b = f"{p}.dyhead_tower.{3 * i + 2}"
        for k in range(3):
            yield f"{b}.DyConv.{k}.conv.weight", (C, C, 3, 3), "w"
            yield f"{b}.DyConv.{k}.conv.bias", (C,), "b"
            yield f"{b}.DyConv.{k}.bn.weight", (C,), "one"
            yield f"{b}.DyConv.{k}.bn.bias", (C,), "b"
        yield b + ".AttnConv.1.weight", (1, C, 1, 1), "w"
        yield b + ".AttnConv.1.bias", (1,), "b"
        yield b + ".relu.fc.0.weight", (C // 4, C), "w"
        yield b + ".relu.fc.0.bias", (C // 4,), "b"
        yield b + ".relu.fc.2.weight", (4 * C, C // 4), "w"
        yield b + ".relu.fc.2.bias", (4 * C,), "b"
        yield b + ".offset.weight", (27, C, 3, 3), "zero"
        yield b + ".offset.bias", (27,), "b"
    ncls = D.NUM_CLASSES - 1
    prior = -math.log((1 - D.PRIOR_PROB) / D.PRIOR_PROB)
    yield p + ".cls_logits.weight", (ncls, C, 1, 1), "w"
    yield p + ".cls_logits.bias", (ncls,), f"const:{prior}"
    yield p + ".bbox_pred.weight", (4, C, 1, 1), "w"
    yield p + ".bbox_pred.bias", (4,), "b"
    yield p + ".centerness.weight", (1, C, 1, 1), "w"
    yield p + ".centerness.bias", (1,), "b"
    yield p + ".dot_product_projection_text.weight", (C, H), "w"
    yield p + ".dot_product_projection_text.bias", (C,), "b"
    yield p + ".log_scale", (1,), f"const:{D.LOG_SCALE}"
    yield p + ".bias_lang", (H,), "b"
    yield p + ".bias0", (1,), f"const:{prior}"
    for l in range(5):
        yield f"{p}.scales.{l}.scale", (1,), "one"
    for l, (s, a) in enumerate(zip(M.RPN.ANCHOR_STRIDE, M.RPN.ANCHOR_SIZES)):
        yield f"rpn.anchor_generator.cell_anchors.{l}", (1, 4), f"buf:anchor:{s}:{a}"


class Container(nn.Module):
    """A bare module that only owns parameters / sub-containers (keeps reference attribute paths alive)."""


def build_param_tree(root, cfg, seed=0, specs=None):
    """Register every parameter of `param_specs` (or the given spec generator) under `root` (an nn.Module) with nested
    Containers.  kind "alias:<name>" registers the SAME Parameter object under a second path (shared modules of the reference:
    state_dict() then lists both names, load_state_dict accepts both)."""
    g = torch.Generator().manual_seed(seed)
    made = {}
    for name, shape, kind in (specs if specs is not None else param_specs(cfg)):
        parts = name.split(".")
        mod = root
        for part in parts[:-1]:
            if part not in mod._modules:
                mod.add_module(part, Container())
            mod = mod._modules[part]
        leaf = parts[-1]
        if kind.startswith("alias:"):
            mod.register_parameter(leaf, made[kind[6:]])
            made[name] = made[kind[6:]]
            continue
        if kind.startswith("buf:"):
            tag = kind.split(":")
            if tag[1] == "relidx":
                mod.register_buffer(leaf, rel_pos_index(int(round(math.sqrt(shape[0])))))
            else:
                mod.register_buffer(leaf, cell_anchor(int(tag[2]), int(tag[3])))
            continue
        if kind == "w":
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            t = torch.randn(*shape, generator=g) * (1.0 / math.sqrt(fan_in))
        elif kind in ("table", "emb"):
            t = torch.randn(*shape, generator=g) * 0.02
        elif kind == "emb1":
            t = torch.randn(*shape, generator=g)
        elif kind == "one":
            t = torch.ones(*shape)
        elif kind.startswith("const:"):
            t = torch.full(shape, float(kind[6:]))
        else:                       # "b" / "zero"
            t = torch.zeros(*shape)
        made[name] = nn.Parameter(t, requires_grad=False)
        mod.register_parameter(leaf, made[name])
    return root


# ------------------------------------------------------------------------------------------------ MQ-GroundingDINO
_GDINO_SWIN = {     # build_swin_transformer's table (groundingdino_new/models/GroundingDINO/backbone/swin_transformer.py:766-784)
    "swin_T_224_1k": dict(EMBED_DIM=96, DEPTHS=(2, 2, 6, 2), NUM_HEADS=(3, 6, 12, 24), WINDOW_SIZE=7),
    "swin_B_224_22k": dict(EMBED_DIM=128, DEPTHS=(2, 2, 18, 2), NUM_HEADS=(4, 8, 16, 32), WINDOW_SIZE=7),
    "swin_B_384_22k": dict(EMBED_DIM=128, DEPTHS=(2, 2, 18, 2), NUM_HEADS=(4, 8, 16, 32), WINDOW_SIZE=12),
    "swin_L_224_22k": dict(EMBED_DIM=192, DEPTHS=(2, 2, 18, 2), NUM_HEADS=(6, 12, 24, 48), WINDOW_SIZE=7),
    "swin_L_384_22k": dict(EMBED_DIM=192, DEPTHS=(2, 2, 18, 2), NUM_HEADS=(6, 12, 24, 48), WINDOW_SIZE=12),
}


def gdino_swin_cfg(cfg):
    """The Swin hyper-parameters GROUNDINGDINO.backbone names, as a node shaped like MODEL.SWINT (what pipeline.swin_forward
    reads).  `GROUNDINGDINO.swin_depths` (not a reference key) overrides the depths -- shallow test models."""
    from ..config import CfgNode
    G = cfg.GROUNDINGDINO
    if G.backbone not in _GDINO_SWIN:
        raise NotImplementedError(f"GROUNDINGDINO.backbone = {G.backbone}: only the Swin backbones are implemented")
    d = dict(_GDINO_SWIN[G.backbone], MLP_RATIO=4)
    if G.get("swin_depths", None):
        d["DEPTHS"] = tuple(G.swin_depths)
    for k in ("FUSED_MLP", "FUSED_MLP_WIDTHS"):
        if k in cfg.MODEL.SWINT:
            d[k] = cfg.MODEL.SWINT[k]
    return CfgNode(d)


def gdino_param_specs(cfg):
    """Parameter names / shapes of the reference's GroundingDINO module tree (groundingdino.py:98-287, transformer.py:40-200,
    fuse_modules.py:99-271, transformer_vanilla.py:65-89, ms_deform_attn.py:136-205), in the order `kind` semantics of
    param_specs.  Names that alias ONE module in the reference (`bbox_embed.i`, `transformer.decoder.bbox_embed.i` with
    dec_pred_bbox_embed_share) are emitted as kind "alias:<first name>"."""
    G = cfg.GROUNDINGDINO
    SW = gdino_swin_cfg(cfg)
    D, F = G.hidden_dim, G.dim_feedforward
    dims = [SW.EMBED_DIM * 2 ** i for i in range(len(SW.DEPTHS))]
    yield from swin_specs(SW, "backbone.0")
    L = G.num_feature_levels
    for l in range(L):
        if l < 3:
            yield f"input_proj.{l}.0.weight", (D, dims[l + 1], 1, 1), "w"
        else:
            yield f"input_proj.{l}.0.weight", (D, dims[3] if l == 3 else D, 3, 3), "w"
        yield f"input_proj.{l}.0.bias", (D,), "b"
        yield f"input_proj.{l}.1.weight", (D,), "one"
        yield f"input_proj.{l}.1.bias", (D,), "b"
    H = cfg.MODEL.LANGUAGE_BACKBONE.LANG_DIM
    yield from language_specs(cfg, "bert", D)
    yield "bert.pooler.dense.weight", (H, H), "w"
    yield "bert.pooler.dense.bias", (H,), "b"
    yield "feat_map.weight", (D, H), "w"
    yield "feat_map.bias", (D,), "b"
    M, P = G.nheads, G.enc_n_points

    def msda(b):
        yield b + ".sampling_offsets.weight", (M * L * P * 2, D), "w"
        yield b + ".sampling_offsets.bias", (M * L * P * 2,), "b"
        yield b + ".attention_weights.weight", (M * L * P, D), "w"
        yield b + ".attention_weights.bias", (M * L * P,), "b"
        for n in ("value_proj", "output_proj"):
            yield f"{b}.{n}.weight", (D, D), "w"
            yield f"{b}.{n}.bias", (D,), "b"

    def mha(b):
        yield b + ".in_proj_weight", (3 * D, D), "w"
        yield b + ".in_proj_bias", (3 * D,), "b"
        yield b + ".out_proj.weight", (D, D), "w"
        yield b + ".out_proj.bias", (D,), "b"

    def norm(b):
        yield b + ".weight", (D,), "one"
        yield b + ".bias", (D,), "b"

    def ffn(b, hid):
        yield b + ".linear1.weight", (hid, D), "w"
        yield b + ".linear1.bias", (hid,), "b"
        yield b + ".linear2.weight", (D, hid), "w"
        yield b + ".linear2.bias", (D,), "b"

    def mlp(b, cin, cout, n, alias=None):
        ws = [cin] + [D] * (n - 1) + [cout]
        for i in range(n):
            kind = f"alias:{alias}.layers.{i}" if alias else None
            yield f"{b}.layers.{i}.weight", (ws[i + 1], ws[i]), (kind + ".weight") if alias else ("zero" if (i == n - 1 and cout == 4) else "w")
            yield f"{b}.layers.{i}.bias", (ws[i + 1],), (kind + ".bias") if alias else "b"
    t = "transformer"
    yield t + ".level_embed", (L, D), "table"
    for i in range(G.enc_layers):
        b = f"{t}.encoder.layers.{i}"
        yield from msda(b + ".self_attn")
        yield from norm(b + ".norm1")
        yield from ffn(b, F)
        yield from norm(b + ".norm2")
    for i in range(G.enc_layers):
        b = f"{t}.encoder.text_layers.{i}"
        yield from mha(b + ".self_attn")
        yield from ffn(b, F // 2)
        yield from norm(b + ".norm1")
        yield from norm(b + ".norm2")
    E = F // 2
    for i in range(G.enc_layers):
        b = f"{t}.encoder.fusion_layers.{i}"
        yield b + ".gamma_v", (D,), "const:0.0001"
        yield b + ".gamma_l", (D,), "const:0.0001"
        yield from norm(b + ".layer_norm_v")
        yield from norm(b + ".layer_norm_l")
        for n, (o, c) in (("v_proj", (E, D)), ("l_proj", (E, D)), ("values_v_proj", (E, D)), ("values_l_proj", (E, D)),
                          ("out_v_proj", (D, E)), ("out_l_proj", (D, E))):
            yield f"{b}.attn.{n}.weight", (o, c), "w"
            yield f"{b}.attn.{n}.bias", (o,), "b"
    for i in range(G.dec_layers):
        b = f"{t}.decoder.layers.{i}"
        yield from msda(b + ".cross_attn")
        yield from norm(b + ".norm1")
        yield from mha(b + ".ca_text")
        yield from norm(b + ".catext_norm")
        yield from mha(b + ".self_attn")
        yield from norm(b + ".norm2")
        yield from ffn(b, F)
        yield from norm(b + ".norm3")
    yield from norm(t + ".decoder.norm")
    yield from mlp(t + ".decoder.ref_point_head", 2 * D, D, 2)
    yield from mlp("bbox_embed.0", D, 4, 3)
    for i in range(G.dec_layers):
        if i > 0:
            yield from mlp(f"bbox_embed.{i}", D, 4, 3, alias="bbox_embed.0" if G.dec_pred_bbox_embed_share else None)
        yield from mlp(f"{t}.decoder.bbox_embed.{i}", D, 4, 3, alias=f"bbox_embed.{i}")
    yield t + ".tgt_embed.weight", (G.num_queries, D), "emb1"
    yield t + ".enc_output.weight", (D, D), "w"
    yield t + ".enc_output.bias", (D,), "b"
    yield from norm(t + ".enc_output_norm")
    yield from mlp(t + ".enc_out_bbox_embed", D, 4, 3, alias="bbox_embed.0" if G.two_stage_bbox_embed_share else None)
