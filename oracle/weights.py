"""Seeded synthetic state_dict with the reference's parameter names and shapes
(test infrastructure, see oracle/__init__.py).

No checkpoints exist in the build or GPU environment, so the oracle and the device model share
this generator (SURVEY.md 8d "Weights").  Names follow SURVEY.md 8b / the reference module
definitions (swint.py, fpn.py, HF BertModel, modeling_bert_new.py, fuse_helper.py,
rpn/modeling_bert.py, vldyhead.py, dyrelu.py, anchor_generator.py).  Values are NOT the reference's
initialisers: every zero-initialised gate / offset / layer-scale parameter is made non-zero so that
all branches are numerically live, and scales are chosen so activations
stay O(1), DCN offsets are O(1-3 px) and a few percent of the alignment scores cross 0.05.
"""
import math

import torch

from .backbone import rel_pos_index
from .postprocess import cell_anchor


class _Gen:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.sd = {}

    def randn(self, *shape, std=1.0, mean=0.0):
        return torch.randn(*shape, generator=self.g) * std + mean

    def linear(self, name, out_f, in_f, bias=True, gain=1.0, bias_std=0.05):
        self.sd[name + ".weight"] = self.randn(out_f, in_f, std=gain / math.sqrt(in_f))
        if bias:
            self.sd[name + ".bias"] = self.randn(out_f, std=bias_std)

    def conv(self, name, out_c, in_c, k, bias=True, gain=1.0, bias_std=0.05):
        self.sd[name + ".weight"] = self.randn(out_c, in_c, k, k, std=gain / math.sqrt(in_c * k * k))
        if bias:
            self.sd[name + ".bias"] = self.randn(out_c, std=bias_std)

    def norm(self, name, c):
        self.sd[name + ".weight"] = self.randn(c, std=0.1, mean=1.0)
        self.sd[name + ".bias"] = self.randn(c, std=0.05)


def _swin(g, p, spec):
    """Swin parameters under prefix `p` (same names in maskrcnn_benchmark's swint.py and GroundingDINO's swin_transformer.py)."""
    sd, ws = g.sd, spec.window
    g.conv(p + ".patch_embed.proj", spec.swin_embed, 3, 4)
    g.norm(p + ".patch_embed.norm", spec.swin_embed)
    for i, (depth, heads) in enumerate(zip(spec.swin_depths, spec.swin_heads)):
        C = spec.swin_dims[i]
        for j in range(depth):
            b = f"{p}.layers.{i}.blocks.{j}"
            g.norm(b + ".norm1", C)
            g.linear(b + ".attn.qkv", 3 * C, C, gain=1.5)
            g.linear(b + ".attn.proj", C, C, gain=0.5)
            sd[b + ".attn.relative_position_bias_table"] = g.randn((2 * ws - 1) ** 2, heads, std=0.5)
            sd[b + ".attn.relative_position_index"] = rel_pos_index(ws)
            g.norm(b + ".norm2", C)
            g.linear(b + ".mlp.fc1", spec.mlp_ratio * C, C)
            g.linear(b + ".mlp.fc2", C, spec.mlp_ratio * C, gain=0.5)
        if i < len(spec.swin_depths) - 1:
            g.norm(f"{p}.layers.{i}.downsample.norm", 4 * C)
            g.linear(f"{p}.layers.{i}.downsample.reduction", 2 * C, 4 * C, bias=False)
        if i > 0:
            g.norm(f"{p}.norm{i}", C)


def _bert_layer(g, b, spec):
    H = spec.bert_hidden
    for n in ("query", "key", "value"):
        g.linear(f"{b}.attention.self.{n}", H, H, gain=1.5 if n != "value" else 1.0)
    g.linear(b + ".attention.output.dense", H, H, gain=0.7)
    g.norm(b + ".attention.output.LayerNorm", H)
    g.linear(b + ".intermediate.dense", spec.bert_inter, H)
    g.linear(b + ".output.dense", H, spec.bert_inter, gain=0.7)
    g.norm(b + ".output.LayerNorm", H)


def _bert_gcp(g, p, spec):
    """BERT-base + GCP (QVBertModel) parameters under prefix `p`."""
    sd = g.sd
    H = spec.bert_hidden
    sd[p + ".embeddings.word_embeddings.weight"] = g.randn(spec.vocab, H, std=0.5)
    sd[p + ".embeddings.position_embeddings.weight"] = g.randn(spec.max_pos, H, std=0.2)
    sd[p + ".embeddings.token_type_embeddings.weight"] = g.randn(2, H, std=0.2)
    g.norm(p + ".embeddings.LayerNorm", H)

    for i in range(spec.bert_layers):
        _bert_layer(g, f"{p}.encoder.layer.{i}", spec)
    if spec.vision_query:
        inner = spec.gcp_heads * spec.gcp_dim_head
        for i in range(spec.bert_layers - spec.qv_start):
            b = f"{p}.encoder.qv_layer.{i}"
            g.norm(b + ".attn.norm", H)
            g.norm(b + ".attn.norm_kv", H)
            g.linear(b + ".attn.to_q", inner, H, bias=False, gain=1.5)
            g.linear(b + ".attn.to_kv", 2 * inner, H, bias=False, gain=1.5)
            g.linear(b + ".attn.to_out", H, inner, bias=False)
            g.norm(b + ".attn_gate.norm", H)
            g.linear(b + ".attn_gate.linear1", H // 2, H, bias=False)
            g.linear(b + ".attn_gate.linear2", 1, H // 2, bias=False, gain=2.0)      # un-zeroed
            g.norm(b + ".ff.norm", H)
            g.linear(b + ".ff.linear1", spec.ff_mult * H, H, bias=False)
            g.linear(b + ".ff.linear2", H, spec.ff_mult * H, bias=False)
            sd[b + ".ff_gate"] = torch.tensor([0.3 + 0.1 * i])                        # un-zeroed
        Cv = spec.fpn_out
        pin = spec.gcp_heads * spec.pre_dim_head
        for i in range(spec.pre_layers):
            b = f"{p}.pre_select.layers.{i}"
            out = Cv if i < spec.pre_layers - 1 else H
            g.norm(b + ".image_condition.norm", Cv)
            g.norm(b + ".image_condition.norm_kv", Cv)
            g.linear(b + ".image_condition.to_q", pin, Cv, bias=False, gain=1.5)
            g.linear(b + ".image_condition.to_kv", 2 * pin, Cv, bias=False, gain=1.5)
            g.linear(b + ".image_condition.to_out", out, pin, bias=False)
            g.norm(b + ".ff.norm", out)
            g.linear(b + ".ff.linear1", spec.ff_mult * out, out, bias=False)
            g.linear(b + ".ff.linear2", out, spec.ff_mult * out, bias=False, gain=0.5)
            if out != Cv:
                g.linear(b + ".res_mapping", out, Cv, bias=False)


def make_state_dict(spec, seed=0):
    g = _Gen(seed)
    sd = g.sd
    _swin(g, "backbone.body", spec)
    # ---------------- FPN
    p = "backbone.fpn"
    dims = spec.swin_dims
    for idx, cin in ((2, dims[1]), (3, dims[2]), (4, dims[3])):
        g.conv(f"{p}.fpn_inner{idx}", spec.fpn_out, cin, 1)
        g.conv(f"{p}.fpn_layer{idx}", spec.fpn_out, spec.fpn_out, 3)
    g.conv(p + ".top_blocks.p6", spec.fpn_out, spec.fpn_out, 3)
    g.conv(p + ".top_blocks.p7", spec.fpn_out, spec.fpn_out, 3)
    _bert_gcp(g, "language_backbone.body.model", spec)
    H = spec.bert_hidden
    # ---------------- VLDyHead
    p = "rpn.head"
    C, E = spec.dyhead_channels, spec.fuse_embed
    for i in range(spec.dyhead_convs):
        b = f"{p}.dyhead_tower.{3 * i}.b_attn"
        g.norm(b + ".layer_norm_v", C)
        g.norm(b + ".layer_norm_l", H)
        g.linear(b + ".attn.v_proj", E, C, gain=1.5)
        g.linear(b + ".attn.l_proj", E, H, gain=1.5)
        g.linear(b + ".attn.values_v_proj", E, C)
        g.linear(b + ".attn.values_l_proj", E, H)
        g.linear(b + ".attn.out_v_proj", C, E)
        g.linear(b + ".attn.out_l_proj", H, E)
        sd[b + ".gamma_v"] = g.randn(C, std=0.05, mean=0.5)
        sd[b + ".gamma_l"] = g.randn(H, std=0.05, mean=0.5)
        _bert_layer(g, f"{p}.dyhead_tower.{3 * i + 1}", spec)
        b = f"{p}.dyhead_tower.{3 * i + 2}"
        for k in range(3):
            g.conv(f"{b}.DyConv.{k}.conv", C, C, 3)
            g.norm(f"{b}.DyConv.{k}.bn", C)
        g.conv(b + ".AttnConv.1", 1, C, 1, gain=4.0, bias_std=0.5)
        g.linear(b + ".relu.fc.0", C // 4, C)
        g.linear(b + ".relu.fc.2", 4 * C, C // 4, gain=2.0, bias_std=0.5)
        g.conv(b + ".offset", 27, C, 3, gain=1.0, bias_std=0.5)                        # offsets ~ O(1 px)
    g.conv(p + ".cls_logits", spec.num_classes - 1, C, 1, gain=0.1)
    sd[p + ".cls_logits.bias"].fill_(-math.log((1 - spec.prior_prob) / spec.prior_prob))
    g.conv(p + ".bbox_pred", 4, C, 1, gain=1.0)
    g.conv(p + ".centerness", 1, C, 1, gain=1.0)
    g.linear(p + ".dot_product_projection_text", C, H, gain=40.0)
    sd[p + ".log_scale"] = torch.tensor([spec.log_scale])
    sd[p + ".bias_lang"] = g.randn(H, std=1.0)
    sd[p + ".bias0"] = torch.tensor([-6.0])
    for l in range(5):
        sd[f"{p}.scales.{l}.scale"] = torch.tensor([1.0 + 0.05 * l])
    for l, (s, a) in enumerate(zip(spec.anchor_strides, spec.anchor_sizes)):
        sd[f"rpn.anchor_generator.cell_anchors.{l}"] = cell_anchor(s, a)
    return sd


def make_query_bank(labels, spec, seed=1, n=None, scales=1):
    """{label: [n, scales, C]} like extract_query's output (generalized_vl_rcnn_new.py:264,285)."""
    g = torch.Generator().manual_seed(seed)
    n = n or spec.num_query_per_class
    return {int(l): torch.randn(n, scales, spec.fpn_out, generator=g) for l in labels}


def make_gdino_state_dict(spec, seed=0):
    """MQ-GroundingDINO parameter names / shapes (groundingdino.py:98-287, transformer.py:40-200, fuse_modules.py:99-271,
    transformer_vanilla.py:65-89, ms_deform_attn.py:136-205, utils.py MLP).  As above, zero-initialised parameters (box-head
    last layer, layer scales, gates) are made non-zero; sampling offsets are O(1-2 px); a common component in the last text
    LayerNorm bias and the (negated) decoder-norm bias centres the contrastive logits around -3 so that the 0.05 box threshold
    (groundingdino.py:303) cuts through the score distribution."""
    g = _Gen(seed)
    sd = g.sd
    D, F = spec.hidden, spec.ffn
    _swin(g, "backbone.0", spec)
    dims = spec.swin_dims
    for l in range(spec.levels):
        if l < 3:
            g.conv(f"input_proj.{l}.0", D, dims[l + 1], 1)
        else:
            g.conv(f"input_proj.{l}.0", D, dims[3] if l == 3 else D, 3)
        g.norm(f"input_proj.{l}.1", D)
    _bert_gcp(g, "bert", spec)
    g.linear("bert.pooler.dense", spec.bert_hidden, spec.bert_hidden)
    g.linear("feat_map", D, spec.bert_hidden)

    def msda(b):
        g.linear(b + ".sampling_offsets", spec.nheads * spec.levels * spec.points * 2, D, gain=1.5, bias_std=1.0)
        g.linear(b + ".attention_weights", spec.nheads * spec.levels * spec.points, D, gain=1.0, bias_std=0.3)
        g.linear(b + ".value_proj", D, D)
        g.linear(b + ".output_proj", D, D, gain=0.7)

    def mha(b):
        sd[b + ".in_proj_weight"] = g.randn(3 * D, D, std=1.3 / math.sqrt(D))
        sd[b + ".in_proj_bias"] = g.randn(3 * D, std=0.05)
        g.linear(b + ".out_proj", D, D, gain=0.7)

    def mlp(b, cin, chid, cout, n, last_gain=1.0):
        ws = [cin] + [chid] * (n - 1) + [cout]
        for i in range(n):
            g.linear(f"{b}.layers.{i}", ws[i + 1], ws[i], gain=last_gain if i == n - 1 else 1.0)

    t = "transformer"
    sd[t + ".level_embed"] = g.randn(spec.levels, D, std=0.3)
    common = torch.ones(D) / math.sqrt(D)                  # unit vector shared by the text and the query side
    for i in range(spec.enc_layers):
        b = f"{t}.encoder.layers.{i}"
        msda(b + ".self_attn")
        g.norm(b + ".norm1", D)
        g.linear(b + ".linear1", F, D)
        g.linear(b + ".linear2", D, F, gain=0.7)
        g.norm(b + ".norm2", D)
        b = f"{t}.encoder.text_layers.{i}"
        mha(b + ".self_attn")
        g.linear(b + ".linear1", F // 2, D)
        g.linear(b + ".linear2", D, F // 2, gain=0.7)
        g.norm(b + ".norm1", D)
        g.norm(b + ".norm2", D)
        if i == spec.enc_layers - 1:
            sd[b + ".norm2.bias"] += 1.0 * common
        b = f"{t}.encoder.fusion_layers.{i}"
        sd[b + ".gamma_v"] = g.randn(D, std=0.05, mean=0.5)
        sd[b + ".gamma_l"] = g.randn(D, std=0.05, mean=0.5)
        g.norm(b + ".layer_norm_v", D)
        g.norm(b + ".layer_norm_l", D)
        E = F // 2
        g.linear(b + ".attn.v_proj", E, D, gain=1.5)
        g.linear(b + ".attn.l_proj", E, D, gain=1.5)
        g.linear(b + ".attn.values_v_proj", E, D)
        g.linear(b + ".attn.values_l_proj", E, D)
        g.linear(b + ".attn.out_v_proj", D, E)
        g.linear(b + ".attn.out_l_proj", D, E)
    for i in range(spec.dec_layers):
        b = f"{t}.decoder.layers.{i}"
        msda(b + ".cross_attn")
        g.norm(b + ".norm1", D)
        mha(b + ".ca_text")
        g.norm(b + ".catext_norm", D)
        mha(b + ".self_attn")
        g.norm(b + ".norm2", D)
        g.linear(b + ".linear1", F, D)
        g.linear(b + ".linear2", D, F, gain=0.7)
        g.norm(b + ".norm3", D)
    sd[t + ".decoder.norm.weight"] = g.randn(D, std=0.02, mean=0.15)
    sd[t + ".decoder.norm.bias"] = g.randn(D, std=0.02) - 3.0 * common
    mlp(t + ".decoder.ref_point_head", 2 * D, D, D, 2)
    mlp("bbox_embed.0", D, D, 4, 3, last_gain=0.3)                      # dec_pred_bbox_embed_share: ONE module, many names
    for i in range(spec.dec_layers):
        for k in [k for k in sd if k.startswith("bbox_embed.0.")]:
            sd[k.replace("bbox_embed.0.", f"bbox_embed.{i}.")] = sd[k]
            sd[t + ".decoder." + k.replace("bbox_embed.0.", f"bbox_embed.{i}.")] = sd[k]
    sd[t + ".tgt_embed.weight"] = g.randn(spec.num_queries, D, std=1.0)
    g.linear(t + ".enc_output", D, D, bias_std=0.0)       # zero bias: masked / invalid tokens score b.text ~ -3, below the live ones
    sd[t + ".enc_output_norm.weight"] = g.randn(D, std=0.02, mean=0.15)
    sd[t + ".enc_output_norm.bias"] = g.randn(D, std=0.02) - 3.0 * common
    mlp(t + ".enc_out_bbox_embed", D, D, 4, 3, last_gain=0.3)
    return sd
