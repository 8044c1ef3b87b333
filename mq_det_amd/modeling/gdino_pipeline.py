"""Functional MI355X forward of MQ-GroundingDINO over an fp16 plan (BASELINE.json configs[4]).

Reference: groundingdino_new/models/GroundingDINO/groundingdino.py:438-661 (GroundingDINO.forward, eval), transformer.py:211-400
(Transformer.forward), :482-596 (encoder: fusion -> text enhancer -> deformable layer), :644-736 and :868-927 (decoder),
fuse_modules.py:146-297 (BiAttentionBlock), transformer_vanilla.py:92-123 (text enhancer layer), ms_deform_attn.py:232-359,
utils.py (proposals, sine embeddings, MLP, ContrastiveEmbed), bertwarper.py:60-215,273-320, backbone/position_encoding.py:76-125.

What runs where (same rules as pipeline.py: fp16 MFMA operands, fp32 accumulation, nothing on the CPU, nothing from the oracle).
Residual streams: the TEXT stream and the DECODER query stream are fp32 end to end.  The IMAGE token stream of the six encoder
layers is fp32 inside a layer (norm1 -> FFN -> norm2 carry `m32`) but crosses the fusion step in 16 bits: mq_vlfuse_i2t_fwd reads
and writes the [B, S, 256] tokens as fp16 / bf16 (like MQ-GLIP's pyramid token buffer), so the residual that enters
norm1 of `deformable_encoder_layer` is the fusion kernel's 16-bit output -- ONE extra rounding of the image stream per layer
(ADVICE r2).  Its weight is measured, not assumed: tests/gdino_checks.py compares the encoder memory after all six layers with the
fp32 oracle (2.7e-3 of the tensor's range at full depth, profiles/r02_gdino_parity.txt), inside the gate of that check:
  * Swin backbone: the MQ-GLIP kernels (window attention, fused MLP, LayerNorm) under the `backbone.0` names;
  * feature-enhancer fusion (4 heads x 256 between ~22 k image tokens and 256 text tokens): the VLFuse kernels with the
    image-side projections folded into the text operands (the 1024-wide image tensors of the reference never exist), the
    image padding mask applied inside the text-side kernel;
  * every nn.MultiheadAttention (text enhancer 4 x 64 with the sub-sentence block mask, decoder self-attention 8 x 32 over
    900 queries, decoder text cross-attention) and the BERT layers: mq_attn_fwd (per-(query, key) byte mask variant);
  * multi-scale deformable attention: mq_msdeform_attn_q_fwd -- softmax over the 16 samples and the sampling locations are
    computed in registers from the fused [offsets | logits] projection, the six decoder layers gather from ONE batched value
    projection of the encoder memory;
  * LayerNorm + residual adds: mq_layernorm_fwd; plain projections / FFNs: library GEMMs.
Geometry that depends only on the padded batch shape and the image sizes (padding masks, sine position embeddings, encoder
reference points, two-stage proposals) is computed once per shape by `geometry()` and cached by the caller.
"""
import math
import os

import torch
import torch.nn.functional as F

from .. import ops
from . import pipeline
from .pipeline import NEG, _add_ln, _lin, _ln, _nsplit, _nsplit_t2i


# ----------------------------------------------------------------------------- plan
def build_gdino_plan(sd, cfg, device, SW, dtype=torch.float16):
    """Pack the fp32 state_dict (reference names) into inference tensors."""
    G = cfg.GROUNDINGDINO
    P = {"_r32": True}

    def h(name):
        return sd[name].detach().to(device=device, dtype=dtype).contiguous()

    def f32(name):
        return sd[name].detach().to(device=device, dtype=torch.float32).contiguous()
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            P[k] = v.detach().to(device=device, dtype=dtype).contiguous()
    pipeline._pack_swin(P, sd, SW, "backbone.0", device, dtype)
    pipeline._pack_language(P, sd, cfg, "bert", device, dtype)
    D, L, nl_e, nl_d = G.hidden_dim, G.num_feature_levels, G.enc_layers, G.dec_layers
    for l in range(L):
        w = f32(f"input_proj.{l}.0.weight")
        if w.shape[-1] == 1:
            P[f"input_proj.{l}.lin"] = w.reshape(w.shape[0], -1).to(dtype).contiguous()
        else:                                                    # 3x3 stride-2 conv: [O, tap*C + c] for mq_conv3x3_fwd
            P[f"input_proj.{l}.packed"] = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(dtype).contiguous()
        P[f"input_proj.{l}.gn.w32"], P[f"input_proj.{l}.gn.b32"] = f32(f"input_proj.{l}.1.weight"), f32(f"input_proj.{l}.1.bias")
    P["level_embed32"] = f32("transformer.level_embed")
    Hf = G.nheads // 2                                           # fusion heads (transformer.py:96-104)
    for i in range(nl_e):
        b = f"transformer.encoder.fusion_layers.{i}"
        E = sd[b + ".attn.v_proj.weight"].shape[0]
        hd = E // Hf
        sc = hd ** -0.5
        gv, gl = f32(b + ".gamma_v"), f32(b + ".gamma_l")
        wq = (f32(b + ".attn.v_proj.weight") * sc).reshape(Hf, hd, -1)                    # [h, e, c]   (scale folded, :176)
        bq = (f32(b + ".attn.v_proj.bias") * sc).reshape(Hf, hd)
        wl, bl = f32(b + ".attn.l_proj.weight").reshape(Hf, hd, -1), f32(b + ".attn.l_proj.bias").reshape(Hf, hd)
        wvl, bvl = f32(b + ".attn.values_l_proj.weight").reshape(Hf, hd, -1), f32(b + ".attn.values_l_proj.bias").reshape(Hf, hd)
        wov = (f32(b + ".attn.out_v_proj.weight") * gv[:, None]).reshape(-1, Hf, hd).permute(1, 2, 0)      # [h, e, c_out]
        # text operands of the layer as ONE projection of LN(l) (see pipeline.build_plan "folded projections"):
        #   Kf_h = (l Wl_h^T + bl_h) Wq_h, Vo_h = (l Wvl_h^T + bvl_h) (gamma_v Wov_h), logit bias (l Wl_h^T + bl_h) . bq_h
        w_kf = torch.einsum("hec,hek->hck", wq, wl).reshape(Hf * wq.shape[2], -1)
        b_kf = torch.einsum("hec,he->hc", wq, bl).reshape(-1)
        w_vo = torch.einsum("heo,hek->hok", wov, wvl).reshape(Hf * wov.shape[2], -1)
        b_vo = torch.einsum("heo,he->ho", wov, bvl).reshape(-1)
        w_b = torch.einsum("he,hek->hk", bq, wl)
        b_b = (bq * bl).sum(-1)
        z = w_b.new_zeros(8 - Hf, w_b.shape[1])
        P[b + ".tprep.weight"] = torch.cat([w_kf, w_vo, w_b, z], 0).to(dtype).contiguous()
        P[b + ".tprep.bias"] = torch.cat([b_kf, b_vo, b_b, b_b.new_zeros(8 - Hf)], 0).to(dtype).contiguous()
        P[b + ".ov.bias"] = (f32(b + ".attn.out_v_proj.bias") * gv).to(dtype)
        # text side: sum_n P_h[t,n] (Wvv_h LN(v)_n + bvv_h) = Wvv_h pooled_h[t] + bvv_h, then out_l_proj and gamma_l -> one weight
        wol = (f32(b + ".attn.out_l_proj.weight") * gl[:, None]).reshape(-1, Hf, hd)      # [o, h, e]
        wvv = f32(b + ".attn.values_v_proj.weight").reshape(Hf, hd, -1)                    # [h, e, c]
        P[b + ".olc.weight"] = torch.einsum("ohe,hec->ohc", wol, wvv).reshape(wol.shape[0], -1).to(dtype).contiguous()
        P[b + ".olc.bias"] = (torch.einsum("ohe,he->o", wol, f32(b + ".attn.values_v_proj.bias").reshape(Hf, hd))
                              + f32(b + ".attn.out_l_proj.bias") * gl).to(dtype)
        P[b + ".heads"] = Hf
        _pack_mha(P, sd, f"transformer.encoder.text_layers.{i}.self_attn", D, h)
        _pack_msda(P, sd, f"transformer.encoder.layers.{i}.self_attn", h)
    for i in range(nl_d):
        b = f"transformer.decoder.layers.{i}"
        _pack_mha(P, sd, b + ".self_attn", D, h)
        _pack_mha(P, sd, b + ".ca_text", D, h)
        _pack_msda(P, sd, b + ".cross_attn", h)
    # the decoder layers all gather from the same memory: ONE value projection [nl_d * D, D] (transformer.py:912-919 x 6)
    P["transformer.decoder.value_all.weight"] = torch.cat([h(f"transformer.decoder.layers.{i}.cross_attn.value_proj.weight")
                                                           for i in range(nl_d)], 0).contiguous()
    P["transformer.decoder.value_all.bias"] = torch.cat([h(f"transformer.decoder.layers.{i}.cross_attn.value_proj.bias")
                                                         for i in range(nl_d)], 0).contiguous()
    # ... and cross-attend to the same text: keys / values of every layer's ca_text in one projection
    P["transformer.decoder.text_k_all.weight"] = torch.cat([P[f"transformer.decoder.layers.{i}.ca_text.k.weight"] for i in range(nl_d)], 0)
    P["transformer.decoder.text_k_all.bias"] = torch.cat([P[f"transformer.decoder.layers.{i}.ca_text.k.bias"] for i in range(nl_d)], 0)
    P["transformer.decoder.text_v_all.weight"] = torch.cat([P[f"transformer.decoder.layers.{i}.ca_text.v.weight"] for i in range(nl_d)], 0)
    P["transformer.decoder.text_v_all.bias"] = torch.cat([P[f"transformer.decoder.layers.{i}.ca_text.v.bias"] for i in range(nl_d)], 0)
    P["tgt_embed32"] = f32("transformer.tgt_embed.weight")
    return P


def _pack_mha(P, sd, b, D, h):
    """nn.MultiheadAttention packed weights -> fused q|k projection + separate q / k / v."""
    W, bias = h(b + ".in_proj_weight"), h(b + ".in_proj_bias")
    P[b + ".qk.weight"], P[b + ".qk.bias"] = W[:2 * D].contiguous(), bias[:2 * D].contiguous()
    P[b + ".q.weight"], P[b + ".q.bias"] = W[:D].contiguous(), bias[:D].contiguous()
    P[b + ".k.weight"], P[b + ".k.bias"] = W[D:2 * D].contiguous(), bias[D:2 * D].contiguous()
    P[b + ".v.weight"], P[b + ".v.bias"] = W[2 * D:].contiguous(), bias[2 * D:].contiguous()


def _pack_msda(P, sd, b, h):
    """[sampling_offsets | attention_weights] as one projection (ms_deform_attn.py:292-298)."""
    P[b + ".qproj.weight"] = torch.cat([h(b + ".sampling_offsets.weight"), h(b + ".attention_weights.weight")], 0).contiguous()
    P[b + ".qproj.bias"] = torch.cat([h(b + ".sampling_offsets.bias"), h(b + ".attention_weights.bias")], 0).contiguous()


# ----------------------------------------------------------------------------- geometry (per padded shape + image sizes)
def level_shapes(H, W, levels):
    """Feature-map sizes of the `levels` pyramid levels for a padded H x W batch: Swin stages 1-3 (patch 4, three 2x merges
    with ceil) and stride-2 3x3 convs after them (groundingdino.py:480-495)."""
    h, w = -(-H // 4), -(-W // 4)
    out = []
    for _ in range(3):
        h, w = (h + 1) // 2, (w + 1) // 2
        out.append((h, w))
    for _ in range(levels - 3):
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        out.append((h, w))
    return out[:levels]


def _sine_hw(mask, num_pos_feats, temperature):
    """PositionEmbeddingSineHW, normalize=True (position_encoding.py:76-125) -> [B, H, W, 2*num_pos_feats]."""
    nm = ~mask
    y = nm.cumsum(1, dtype=torch.float32)
    x = nm.cumsum(2, dtype=torch.float32)
    y = y / (y[:, -1:, :] + 1e-6) * (2 * math.pi)
    x = x / (x[:, :, -1:] + 1e-6) * (2 * math.pi)
    d = torch.arange(num_pos_feats, dtype=torch.float32, device=mask.device)
    d = temperature ** (2 * torch.div(d, 2, rounding_mode="floor") / num_pos_feats)
    px, py = x[..., None] / d, y[..., None] / d
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), 4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), 4).flatten(3)
    return torch.cat((py, px), 3)


def geometry(P, cfg, H, W, image_sizes, device):
    """Everything the transformer needs that depends only on the padded batch shape and the (h, w) of every image."""
    G = cfg.GROUNDINGDINO
    L, D = G.num_feature_levels, G.hidden_dim
    B = len(image_sizes)
    full = torch.zeros(B, H, W, dtype=torch.bool, device=device)             # util/misc.py:474-487: True = padding
    for i, (h, w) in enumerate(image_sizes):
        full[i, int(h):, :] = True
        full[i, :, int(w):] = True
    shapes = level_shapes(H, W, L)
    masks = [F.interpolate(full[None].float(), size=s).to(torch.bool)[0] for s in shapes]      # nearest (backbone.py:141-143)
    mask = torch.cat([m.flatten(1) for m in masks], 1)                                        # [B, S]
    pos = torch.cat([(_sine_hw(m, D // 2, float(G.pe_temperatureH)) + P["level_embed32"][l]).flatten(1, 2)
                     for l, m in enumerate(masks)], 1)                                        # [B, S, D] incl. level embedding
    vr = torch.stack([torch.stack([(~m[:, 0, :]).sum(1).float() / m.shape[2], (~m[:, :, 0]).sum(1).float() / m.shape[1]], -1)
                      for m in masks], 1)                                                     # [B, L, 2] (w, h)  transformer.py:190-197
    refs, props = [], []
    for lvl, ((h, w), m) in enumerate(zip(shapes, masks)):
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=device),
                                torch.arange(w, dtype=torch.float32, device=device), indexing="ij")
        # encoder reference points (transformer.py:467-481)
        ry = (ys + 0.5).reshape(-1)[None] / (vr[:, None, lvl, 1] * h)
        rx = (xs + 0.5).reshape(-1)[None] / (vr[:, None, lvl, 0] * w)
        refs.append(torch.stack((rx, ry), -1))
        # two-stage proposals (utils.py:52-108): centres normalised by the VALID extent, wh = 0.05 * 2^lvl
        vh, vw = (~m[:, :, 0]).sum(1), (~m[:, 0, :]).sum(1)
        scale = torch.stack([vw, vh], 1).view(B, 1, 1, 2).float()
        grid = (torch.stack([xs, ys], -1)[None].expand(B, -1, -1, -1) + 0.5) / scale
        props.append(torch.cat((grid, torch.ones_like(grid) * 0.05 * (2.0 ** lvl)), -1).view(B, -1, 4))
    ref = (torch.cat(refs, 1)[:, :, None] * vr[:, None]).contiguous()                          # [B, S, L, 2]
    props = torch.cat(props, 1)
    valid = ((props > 0.01) & (props < 0.99)).all(-1)
    invalid = mask | ~valid
    props = torch.log(props / (1 - props)).masked_fill(invalid[..., None], float("inf"))
    valid_hw = torch.stack([torch.stack([(~m[:, :, 0]).sum(1), (~m[:, 0, :]).sum(1)], -1) for m in masks], 1).to(torch.int32)
    return {"shapes": tuple(shapes), "mask": mask, "key_mask": ops.image_key_mask(mask), "pos": pos.contiguous(), "vr": vr,
            "pos16": pos.to(P["transformer.level_embed"].dtype).contiguous(), "valid_hw": valid_hw.contiguous(),
            "enc_ref": ref, "proposals": props.contiguous(), "invalid": invalid, "any_pad": bool(mask.any())}


# ----------------------------------------------------------------------------- text-side host preparation
def special_token_masks(input_ids, special_ids):
    """bertwarper.py:273-320 on the host: block self-attention mask [B,T,T] (True = may attend) + per-block position ids."""
    B, T = input_ids.shape
    special = torch.zeros(B, T, dtype=torch.bool)
    for s in special_ids:
        special |= input_ids == s
    attn = torch.eye(T, dtype=torch.bool)[None].repeat(B, 1, 1)
    pos = torch.zeros(B, T, dtype=torch.long)
    prev = 0                                                   # carried across rows, like the reference's loop
    for row, col in torch.nonzero(special).tolist():
        if col == 0 or col == T - 1:
            attn[row, col, col] = True
            pos[row, col] = 0
        else:
            attn[row, prev + 1:col + 1, prev + 1:col + 1] = True
            pos[row, prev + 1:col + 1] = torch.arange(0, col - prev)
        prev = col
    return attn, pos


def sine_pos_embed(pos, num_pos_feats, temperature=10000):
    """utils.py get_sine_pos_embed(exchange_xy=False) for one coordinate: [B,T] -> [B,T,num_pos_feats]."""
    d = torch.arange(num_pos_feats, dtype=torch.float32, device=pos.device)
    d = temperature ** (2 * torch.div(d, 2, rounding_mode="floor") / num_pos_feats)
    s = pos[..., None].float() * (2 * math.pi) / d
    return torch.stack((s[..., 0::2].sin(), s[..., 1::2].cos()), 3).flatten(2)


def text_inputs(cfg, input_ids_full, attention_mask_full, special_ids, device):
    """Host part of groundingdino.py:518-571: sub-sentence masks on the tokenizer's full-width output, cut to max_text_len.
    Returns (dict of device tensors, max_kv = host-side length of the longest caption in tokens)."""
    G = cfg.GROUNDINGDINO
    T = G.max_text_len
    attn, pos = special_token_masks(input_ids_full.cpu(), special_ids)
    attn, pos = attn[:, :T, :T], pos[:, :T]
    ids, am = input_ids_full[:, :T].cpu(), attention_mask_full[:, :T].cpu().bool()
    B = ids.shape[0]
    hidden = (~attn).to(torch.uint8).to(device)                                 # 1 = key hidden from query
    He = G.nheads // 2
    if B > 1 and not bool((attn == attn[:1]).all()):
        # transformer_vanilla.py:108-109 repeats the [B,T,T] mask `nhead` times along dim 0 and nn.MultiheadAttention reads it
        # as [b*nhead + h]: head (b, h) of the text enhancer uses the mask of batch element (b*nhead + h) % B
        sel = ((torch.arange(B)[:, None] * He + torch.arange(He)[None]) % B).to(device)
        enh = hidden[sel].contiguous()                                          # [B, He, T, T]
    else:
        enh = hidden[:, None].expand(B, He, T, T)
    kv_len = (am.to(torch.int32) * torch.arange(1, T + 1, dtype=torch.int32)).amax(1).to(torch.int32)
    txt = {"input_ids": ids.to(device), "position_ids": pos.to(device), "token_mask": am.to(device), "bert_mask": hidden[:, None],
           "enh_mask": enh, "key_bias": ((~am).float() * NEG).to(device).contiguous(), "kv_len": kv_len.to(device).contiguous(),
           "pos_text": sine_pos_embed(pos, 256).to(device)}
    return txt, int((am.long() * torch.arange(1, T + 1)).max())


# ----------------------------------------------------------------------------- blocks
def group_norm_tokens(x, w32, b32, groups, eps=1e-5):
    """nn.GroupNorm(groups, C) on channel-last tokens [B, N, C] -> fp32 (input_proj, groundingdino.py:200-226)."""
    B, N, C = x.shape
    xf = x.float().view(B, N, groups, C // groups)
    mean = xf.mean((1, 3), keepdim=True)
    var = xf.var((1, 3), unbiased=False, keepdim=True)
    return ((xf - mean) * torch.rsqrt(var + eps)).view(B, N, C) * w32 + b32


def input_projections(P, cfg, feats):
    """Swin stages 1-3 (NHWC) -> the `num_feature_levels` projected levels as one token buffer [B, S, 256] fp32."""
    G = cfg.GROUNDINGDINO
    toks = []
    for l in range(G.num_feature_levels):
        if l < len(feats):
            B, h, w, C = feats[l].shape
            y = F.linear(feats[l].reshape(B, h * w, C), P[f"input_proj.{l}.lin"], P[f"input_proj.{l}.0.bias"])
        else:
            src = feats[-1] if l == len(feats) else prev
            y = ops.conv3x3(src.contiguous(), P[f"input_proj.{l}.packed"], P[f"input_proj.{l}.0.bias"], G.hidden_dim, stride=2)
            B, h, w, _ = y.shape
            y = y.reshape(B, h * w, -1)
        t = group_norm_tokens(y, P[f"input_proj.{l}.gn.w32"], P[f"input_proj.{l}.gn.b32"], 32)
        prev = t.to(y.dtype).view(B, h, w, -1)
        toks.append(t)
    return torch.cat(toks, 1)


def _vt(P, name, x):
    """V^T = W x^T + b for the attention kernel: x [B,N,C] -> [B,C,N_pad8]."""
    B, N, _ = x.shape
    pad = (-N) % 8
    if pad:
        x = F.pad(x, (0, 0, 0, pad))
    return torch.baddbmm(P[name + ".bias"][None, :, None], P[name + ".weight"][None].expand(B, -1, -1), x.transpose(1, 2))


def fusion_layer(P, b, mem32, text32, geo, txt, max_kv=0):
    """BiAttentionBlock (fuse_modules.py:252-297): returns (image tokens fp16 = LN(v) + gamma_v * delta_v, text fp32)."""
    Hf = P[b + ".heads"]
    v_ln = _ln(P, b + ".layer_norm_v", mem32)
    l16, l32 = _ln(P, b + ".layer_norm_l", text32, want_y32=True)
    B, T, _ = l16.shape
    hd = 256
    pr = _lin(P, b + ".tprep", l16)
    n = Hf * hd
    kf = pr[..., :n].unflatten(-1, (Hf, hd)).permute(0, 2, 1, 3)             # views of the projection output: the kernels take the strides
    vo = pr[..., n:2 * n].unflatten(-1, (Hf, hd)).permute(0, 2, 1, 3)
    bias = (pr[..., 2 * n:2 * n + Hf].float().permute(0, 2, 1) + txt["key_bias"][:, None, :]).contiguous()
    img = ops.vlfuse_i2t(v_ln, kf, vo, bias, P[b + ".ov.bias"], txt["kv_len"], max_kv)
    N = v_ln.shape[1]
    t_live = min(T, max_kv) if (txt["kv_len"] is not None and max_kv > 0) else T
    pooled = ops.vlfuse_t2i(kf, v_ln, _nsplit_t2i(B, Hf, t_live, -(-N // 64)), kv_len=txt["kv_len"],
                            key_mask=geo["key_mask"] if geo["any_pad"] else None, max_kv=t_live)
    return img, l32 + _lin(P, b + ".olc", pooled).float()


def text_enhancer_layer(P, b, text32, txt, heads):
    """transformer_vanilla.py:92-123: q = k = x + pos, v = x, block mask, post-norm, ReLU FFN."""
    t16 = text32.to(P[b + ".norm1.weight"].dtype)
    qk = _lin(P, b + ".self_attn.qk", (text32 + txt["pos_text"]).to(t16.dtype))
    C = t16.shape[-1]
    ctx = ops.attention(qk[..., :C], qk[..., C:], _vt(P, b + ".self_attn.v", t16), heads, C // heads, qk_mask=txt["enh_mask"])
    a16, a32 = _add_ln(P, b + ".norm1", _lin(P, b + ".self_attn.out_proj", ctx), text32, want_sum=False, want_y32=True)
    h = _lin(P, b + ".linear2", F.relu(_lin(P, b + ".linear1", a16)))
    return _add_ln(P, b + ".norm2", h, a32, want_sum=False, want_y32=True)[1]


def deformable_encoder_layer(P, b, mem16, geo, heads):
    """transformer.py:739-804: MSDeformAttn(query = src + pos, value = src) + FFN, post-norm.  mem16: fp16 tokens."""
    qp = _lin(P, b + ".self_attn.qproj", mem16 + geo["pos16"])
    val = _lin(P, b + ".self_attn.value_proj", mem16)
    a = ops.ms_deform_attn_q(val, geo["shapes"], qp, geo["enc_ref"], heads, valid_hw=geo["valid_hw"] if geo["any_pad"] else None)
    m16, m32 = _add_ln(P, b + ".norm1", _lin(P, b + ".self_attn.output_proj", a), mem16, want_sum=False, want_y32=True)
    _mark("enc.deformable")
    h = _lin(P, b + ".linear2", _lin_relu(P, b + ".linear1", m16))
    return _add_ln(P, b + ".norm2", h, m32, want_sum=False, want_y32=True)


_FUSED_RELU = os.environ.get("MQ_GDINO_FUSED_RELU", "1") == "1"


def _lin_relu(P, name, x):
    """relu(x W^T + b): the ReLU rides in the library GEMM's epilogue (hipBLASLt) -- the [B, S, 2048] hidden activation of the
    encoder FFN is written once instead of written, read and written again."""
    if not _FUSED_RELU:
        return F.relu(_lin(P, name, x))
    w = P[name + ".weight"]
    return torch._addmm_activation(P[name + ".bias"], x.reshape(-1, x.shape[-1]), w.t()).view(*x.shape[:-1], w.shape[0])


def _mlp(P, b, x, n):
    for i in range(n):
        x = _lin(P, f"{b}.layers.{i}", x)
        if i < n - 1:
            x = F.relu(x)
    return x


def inverse_sigmoid(x, eps=1e-3):
    x = x.clamp(0, 1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


_SINE = {}


def _box_sine(box):
    """utils.py gen_sineembed_for_position, 4-d: [B, nq, 4] (cx, cy, w, h) -> [B, nq, 512] in (y, x, w, h) order.  The reference
    interleaves sin of the even and cos of the odd frequency slots (a dozen small kernels per coordinate); cos(a) = sin(a + pi/2)
    turns the whole embedding into ONE sine of an affine map: coordinate gather, multiply-add, sin."""
    key = box.device
    if key not in _SINE:
        j = torch.arange(128, dtype=torch.float32, device=box.device)
        freq = (2 * math.pi) / (10000 ** (2 * torch.div(j, 2, rounding_mode="floor") / 128))
        _SINE[key] = (freq, (j % 2) * (math.pi / 2), torch.tensor([1, 0, 2, 3], device=box.device))
    freq, phase, order = _SINE[key]
    return torch.sin(torch.addcmul(phase, box[..., order, None], freq)).flatten(2)


# ----------------------------------------------------------------------------- language
def language(P, cfg, txt, vision, images, idx, want_gates=False, front=None):
    """BertModelWarper.forward over QVBertModel (bertwarper.py:60-215): embeddings with per-sub-sentence positions, pre-select,
    GCP blocks + BERT layers under the block mask -> last hidden state (fp16 operand, fp32 stream)."""
    p = "bert"
    LB = cfg.MODEL.LANGUAGE_BACKBONE
    use_vq = vision is not None
    if front is None:
        front = language_front(P, cfg, txt, use_vq)
    x, x32 = front["x"], front["x32"]
    if use_vq:
        vision = pipeline.pre_select(P, p + ".pre_select", vision, images, cfg.VISION_QUERY.VISION_SCALE)
    nl, qv0 = LB.get("NUM_HIDDEN_LAYERS", 12), LB.get("QV_START", 6)
    gates = [] if want_gates else None
    for i in range(front["next"], nl):
        if use_vq and i >= qv0:
            x32 = pipeline.gcp_block(P, f"{p}.encoder.qv_layer.{i - qv0}", x32, vision, idx, gates)
            x = x32.to(x.dtype)
        x, x32 = pipeline._bert(P, f"{p}.encoder.layer.{i}", x, x32, None, False, None, txt["bert_mask"].expand(-1, 12, -1, -1))
    return x, x32, gates


def language_front(P, cfg, txt, use_vq):
    mask = txt["bert_mask"].expand(-1, 12, -1, -1)
    return pipeline.language_front(P, cfg, txt["input_ids"], None, use_vq, p="bert", position_ids=txt["position_ids"], qk_mask=mask)


_MARKS = None           # bench.py stage timing (eager passes only): list of (stage name, event) or None


def start_marks():
    global _MARKS
    _MARKS = []


def stop_marks():
    """-> {stage: ms} summed over the recorded passes (time between consecutive marks is charged to the later mark's name)."""
    global _MARKS
    marks, _MARKS = _MARKS, None
    torch.cuda.synchronize()
    out = {}
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        if n1 != "start":
            out[n1] = out.get(n1, 0.0) + e0.elapsed_time(e1)
    return out


def _mark(name):
    if _MARKS is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        _MARKS.append((name, e))


# ----------------------------------------------------------------------------- transformer
def encoder(P, cfg, src32, text32, geo, txt, max_kv=0, trace=None):
    G = cfg.GROUNDINGDINO
    mem32 = src32
    for i in range(G.enc_layers):
        t = "transformer.encoder"
        mem16, text32 = fusion_layer(P, f"{t}.fusion_layers.{i}", mem32, text32, geo, txt, max_kv)
        _mark("enc.fusion")
        text32 = text_enhancer_layer(P, f"{t}.text_layers.{i}", text32, txt, G.nheads // 2)
        _mark("enc.text_layer")
        mem16, mem32 = deformable_encoder_layer(P, f"{t}.layers.{i}", mem16, geo, G.nheads)
        _mark("enc.deformable+ffn")
        if trace is not None:
            trace.append({"memory": mem32, "text": text32})
    return mem16, mem32, text32


def two_stage(P, cfg, mem16, text32, geo, txt):
    """transformer.py:262-306: proposals from the encoder memory, top-k by the best token logit; the box head runs on the
    selected rows only (a row-wise MLP: identical to selecting from the full result).  mem16: the fp16 operand copy of the
    encoder output (what the enc_output GEMM consumes anyway)."""
    G = cfg.GROUNDINGDINO
    t = "transformer"
    omem = mem16.masked_fill(geo["invalid"][..., None], 0.0)
    e16, e32 = _ln(P, t + ".enc_output_norm", _lin(P, t + ".enc_output", omem), want_y32=True)
    # fp32 logits (they rank 900 of ~22 k rows); the text padding mask rides in the GEMM as its additive term (-1e30 columns)
    B, S = e32.shape[:2]
    logits = torch.baddbmm(txt["key_bias"][:, None, :].expand(B, S, -1), e32, text32.transpose(1, 2))
    topk = torch.topk(logits.amax(-1), G.num_queries, dim=1)[1]                               # [B, nq]
    sel16 = torch.gather(e16, 1, topk[..., None].expand(-1, -1, e16.shape[-1]))
    props = torch.gather(geo["proposals"], 1, topk[..., None].expand(-1, -1, 4))
    ref_unsig = _mlp(P, t + ".enc_out_bbox_embed", sel16, 3).float() + props
    return ref_unsig.sigmoid(), topk, torch.gather(e32, 1, topk[..., None].expand(-1, -1, e32.shape[-1])), props.sigmoid()


def decoder(P, cfg, mem16, text32, ref0, geo, txt, trace=None):
    """TransformerDecoder.forward (transformer.py:644-736) + DeformableTransformerDecoderLayer (:868-927), batch-first.
    Returns (hs_last fp32 = norm(output of the last layer), refs: list of sigmoid boxes, refs[i] = input of layer i)."""
    G = cfg.GROUNDINGDINO
    t = "transformer.decoder"
    D, M, nl = G.hidden_dim, G.nheads, G.dec_layers
    B = mem16.shape[0]
    dt = mem16.dtype
    val_all = _lin(P, t + ".value_all", mem16)                                                # [B, S, nl*D]
    valid_hw = geo["valid_hw"] if geo["any_pad"] else None
    text16 = text32.to(dt)
    tk_all = _lin(P, t + ".text_k_all", text16)                                               # [B, T, nl*D]
    tvt_all = _vt(P, t + ".text_v_all", text16)                                               # [B, nl*D, T]
    tgt32 = P["tgt_embed32"][None].expand(B, -1, -1).contiguous()
    vr4 = torch.cat([geo["vr"], geo["vr"]], -1)[:, None]                                      # [B, 1, L, 4]
    rp, refs, hs = ref0, [ref0], None
    for i in range(nl):
        b = f"{t}.layers.{i}"
        ref_in = (rp[:, :, None] * vr4).contiguous()                                          # [B, nq, L, 4]
        qpos = _mlp(P, t + ".ref_point_head", _box_sine(ref_in[:, :, 0, :]).to(dt), 2).float()
        # self-attention among the queries
        t16 = tgt32.to(dt)
        qk = _lin(P, b + ".self_attn.qk", (tgt32 + qpos).to(dt))
        ctx = ops.attention(qk[..., :D], qk[..., D:], _vt(P, b + ".self_attn.v", t16), M, D // M, nk=t16.shape[1])
        _, tgt32 = _add_ln(P, b + ".norm2", _lin(P, b + ".self_attn.out_proj", ctx), tgt32, want_sum=False, want_y32=True)
        # text cross-attention
        q = _lin(P, b + ".ca_text.q", (tgt32 + qpos).to(dt))
        ctx = ops.attention(q, tk_all[..., i * D:(i + 1) * D], tvt_all[:, i * D:(i + 1) * D], M, D // M, key_bias=txt["key_bias"],
                            kv_len=txt["kv_len"])
        _, tgt32 = _add_ln(P, b + ".catext_norm", _lin(P, b + ".ca_text.out_proj", ctx), tgt32, want_sum=False, want_y32=True)
        # deformable cross-attention into the encoder memory
        qp = _lin(P, b + ".cross_attn.qproj", (tgt32 + qpos).to(dt))
        a = ops.ms_deform_attn_q(val_all[..., i * D:(i + 1) * D], geo["shapes"], qp, ref_in, M, valid_hw=valid_hw)
        t16, tgt32 = _add_ln(P, b + ".norm1", _lin(P, b + ".cross_attn.output_proj", a), tgt32, want_sum=False, want_y32=True)
        hmid = _lin(P, b + ".linear2", F.relu(_lin(P, b + ".linear1", t16)))
        t16, tgt32 = _add_ln(P, b + ".norm3", hmid, tgt32, want_sum=False, want_y32=True)
        if trace is not None:
            trace.append({"tgt": tgt32, "ref_in": rp})
        if i < nl - 1:                                         # iterative box refinement (:716-727); the last update is unused
            rp = (_mlp(P, f"{t}.bbox_embed.{i}", t16, 3).float() + inverse_sigmoid(rp)).sigmoid()
            refs.append(rp)
        else:
            hs = _ln(P, t + ".norm", tgt32, want_y32=True)
    return hs, refs


def heads(P, cfg, hs, ref_last, text32, txt):
    """groundingdino.py:585-604,641: boxes of the last decoder layer and sigmoid token scores [B, nq, T] (padding tokens 0)."""
    G = cfg.GROUNDINGDINO
    hs16, hs32 = hs
    boxes = (_mlp(P, f"bbox_embed.{G.dec_layers - 1}", hs16, 3).float() + inverse_sigmoid(ref_last)).sigmoid()
    logits = torch.matmul(hs32, text32.transpose(1, 2)).masked_fill(~txt["token_mask"][:, None, :], float("-inf"))
    return logits.sigmoid(), boxes


def convert(prob, boxes, class_map, nan_labels, im_hw, box_threshold):
    """convert_groundingdino_to_glip_output (groundingdino.py:291-335) as fixed-shape device work: class score = mean of the
    class's token scores (class_map [T, C] holds 1/len), best class per query, threshold, cxcywh -> xyxy in pixels, clip to
    [0, W-1] x [0, H-1].  Returns (packed [B, nq, 6] = x1 y1 x2 y2 score label, keep [B, nq] bool)."""
    scores = torch.matmul(prob, class_map)                                                    # [B, nq, C]
    sc, lab = scores.max(-1)
    keep = sc > box_threshold
    if nan_labels:           # a label with an empty token list: NaN class score -> NaN max -> no query passes (reference quirk)
        keep = torch.zeros_like(keep)
    H, W = im_hw[:, 0:1], im_hw[:, 1:2]                                                       # [B, 1]
    whwh = torch.stack([W, H, W, H], -1)                                                      # [B, 1, 4]
    bx = boxes * whwh
    xy1 = bx[..., :2] - bx[..., 2:] / 2
    xy2 = bx[..., 2:] + xy1
    x1, y1 = xy1[..., 0].clamp(min=0), xy1[..., 1].clamp(min=0)
    x1, y1 = torch.minimum(x1, W - 1), torch.minimum(y1, H - 1)
    x2, y2 = torch.minimum(xy2[..., 0].clamp(min=0), W - 1), torch.minimum(xy2[..., 1].clamp(min=0), H - 1)
    # queries below the threshold carry score -1: the fixed-shape tensor can then cross ranks as it is
    # (parallel.gather_detections / unpack_detections keep the rows with score > 0)
    packed = torch.stack([x1, y1, x2, y2, torch.where(keep, sc, torch.full_like(sc, -1.0)), (lab + 1).float()], -1)
    return packed, keep


def forward_device(P, cfg, SW, x, geo, txt, vision, idx, class_map, im_hw, max_kv=0, nan_labels=False, front=None, trace=None,
                   src32=None):
    """The device program: pixels [B,3,H,W] fp16 (channels_last) -> packed detections.  No host synchronisation inside.
    src32 [B, S, 256] fp32: the projected levels of a previous call with the same pixels (the evaluation loop sends every image
    once per chunk caption, engine/inference.py:605-625) -- Swin and the input projections are then skipped, `x` is ignored."""
    G = cfg.GROUNDINGDINO
    if src32 is not None:
        return _forward_from_src(P, cfg, src32, P["backbone.0.patch_embed.proj.weight"].dtype, geo, txt, vision, idx, class_map, im_hw,
                                 max_kv, nan_labels, front, trace)
    _mark("start")
    # the BERT layers below the first GCP block do not depend on the image: on a side stream under the Swin backbone
    # (B x 256 tokens per launch: they would otherwise run alone on a nearly empty chip)
    side = None
    if front is None and x.is_cuda and _MARKS is None and G.get("text_stream", True):
        main, side = torch.cuda.current_stream(), pipeline._side_streams(x.device, 1, "text")[0]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            front = language_front(P, cfg, txt, vision is not None)
    feats = pipeline.swin_forward(P, cfg, x, p="backbone.0", SW=SW)
    if side is not None:
        main.wait_stream(side)
    _mark("swin")
    src32 = input_projections(P, cfg, feats)
    _mark("input_proj")
    return _forward_from_src(P, cfg, src32, x.dtype, geo, txt, vision, idx, class_map, im_hw, max_kv, nan_labels, front, trace)


def _forward_from_src(P, cfg, src32, dt, geo, txt, vision, idx, class_map, im_hw, max_kv, nan_labels, front, trace):
    G = cfg.GROUNDINGDINO
    images = None
    if vision is not None:                                     # flatten_fpn_features (groundingdino.py:423-425)
        views, s0 = [], 0
        src16 = src32.to(dt)
        for (h, w) in geo["shapes"]:
            views.append(src16[:, s0:s0 + h * w].reshape(-1, h, w, src16.shape[-1]).permute(0, 3, 1, 2))
            s0 += h * w
        images = pipeline.pooled_fpn_tokens(views)
    x16, x32, gates = language(P, cfg, txt, vision, images, idx, cfg.VISION_QUERY.RETURN_ATTN_GATE_VALUE, front)
    text32 = _lin(P, "feat_map", x16).float()
    _mark("language")
    if trace is not None:
        trace.update(srcs=src32, bert=x32, encoded_text=text32, enc=[], dec=[])
    mem16, mem32, text32 = encoder(P, cfg, src32, text32, geo, txt, max_kv, None if trace is None else trace["enc"])
    ref0, topk, hs_enc, init_box = two_stage(P, cfg, mem16, text32, geo, txt)
    _mark("two_stage")
    hs, refs = decoder(P, cfg, mem16, text32, ref0, geo, txt, None if trace is None else trace["dec"])
    _mark("decoder")
    prob, boxes = heads(P, cfg, hs, refs[-1], text32, txt)
    packed, keep = convert(prob, boxes, class_map, nan_labels, im_hw, float(G.box_threshold))
    _mark("heads+convert")
    out = {"packed": packed, "keep": keep, "srcs": src32}
    if gates is not None:
        out["gates"] = torch.stack([g.float().mean() for g in gates])
    if trace is not None:
        trace.update(memory=mem32, memory_text=text32, topk=topk, hs_enc=hs_enc, init_box=init_box, refs=refs, hs=hs[1],
                     pred_logits=prob, pred_boxes=boxes)
    return out
