"""Oracle: VLDyHead = 6 x {VLFuse (bi-directional image<->text MHA), clamped BERT layer, DyConv}
+ box / centerness / dot-product alignment heads (test infrastructure, see oracle/__init__.py).

Restates
  * maskrcnn_benchmark/utils/fuse_helper.py:218-303 (BiMultiHeadAttention), :377-426
    (BiAttentionBlockForCheckpoint incl. the residual-on-LayerNormed-input quirk),
  * maskrcnn_benchmark/csrc/cuda/deform_conv_kernel_cuda.cu:475-503,578-640 and
    deform_conv_cuda.cu:496-575 (DCNv2 forward; offsets/masks indexed FLAT by the OUTPUT dims),
  * maskrcnn_benchmark/layers/dyrelu.py:78-112 (DYReLU, exp=4, no spatial branch),
  * maskrcnn_benchmark/modeling/rpn/vldyhead.py:42-49 (h_sigmoid), :205-247 (DyConv),
    :264-301 (BertEncoderLayer), :769-900 (VLDyHead.forward).
"""
import math

import torch
import torch.nn.functional as F

from .language import bert_layer, extended_mask


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


# --------------------------------------------------------------------------- VLFuse
def bi_multihead_attention(sd, p, v, l, mask_l, heads, embed):
    """fuse_helper.py:218-303.  v: [B, N, Cv] image tokens, l: [B, T, Cl] text, mask_l: [B, T] int.
    One logits matrix, two softmaxes: over T (with text mask) for the image side, over N (NO image
    padding mask, taken before the text mask is added) for the text side.  Clamps at +-50000."""
    B, N, _ = v.shape
    T = l.shape[1]
    hd = embed // heads

    def split(t, n):
        return t.reshape(B, n, heads, hd).transpose(1, 2)               # [B, h, n, hd]
    q = split(_lin(sd, p + ".v_proj", v) * hd ** -0.5, N)
    k = split(_lin(sd, p + ".l_proj", l), T)
    val_v = split(_lin(sd, p + ".values_v_proj", v), N)
    val_l = split(_lin(sd, p + ".values_l_proj", l), T)
    w = (q @ k.transpose(-1, -2)).clamp(min=-50000, max=50000)          # [B, h, N, T]
    wt = w.transpose(-1, -2)
    wl = (wt - wt.max(-1, keepdim=True)[0]).clamp(min=-50000, max=50000).softmax(-1)   # [B, h, T, N]
    if mask_l is not None:
        # int64 mask: masked_fill(==0, -9e15) keeps +1 at valid positions (fuse_helper.py:262-270)
        add = mask_l[:, None, None, :].expand(B, 1, N, T)
        add = add.masked_fill(add == 0, -9e15)
        w = w + add
    wv = w.softmax(-1)
    out_v = (wv @ val_l).transpose(1, 2).reshape(B, N, embed)
    out_l = (wl @ val_v).transpose(1, 2).reshape(B, T, embed)
    return _lin(sd, p + ".out_v_proj", out_v), _lin(sd, p + ".out_l_proj", out_l)


def vl_fuse(sd, p, feats, lang_hidden, mask_l, spec):
    """fuse_helper.py:377-426 (MHA-B, not SEPARATE_BIDIRECTIONAL).  `p` = '...dyhead_tower.{3i}.b_attn'."""
    B = feats[0].shape[0]
    sizes = [f.shape[-2:] for f in feats]
    v = torch.cat([f.flatten(2).transpose(1, 2) for f in feats], 1)      # [B, N, C]
    v = _ln(sd, p + ".layer_norm_v", v)
    l = _ln(sd, p + ".layer_norm_l", lang_hidden)
    dv, dl = bi_multihead_attention(sd, p + ".attn", v, l, mask_l, spec.fuse_heads, spec.fuse_embed)
    v = v + sd[p + ".gamma_v"] * dv                                      # residual on the NORMED v,l
    l = l + sd[p + ".gamma_l"] * dl
    v = v.transpose(1, 2)
    out, s = [], 0
    for (h, w) in sizes:
        out.append(v[:, :, s:s + h * w].reshape(B, -1, h, w).contiguous())
        s += h * w
    return out, l


# --------------------------------------------------------------------------- DCNv2
def bilinear_zero(x, h, w):
    """dmcn_im2col_bilinear (deform_conv_kernel_cuda.cu:475-503) with the caller's range test
    (:622): zero outside (-1, H) x (-1, W), per-corner zero padding inside.
    x: [B, C, H, W]; h, w: [B, n] float sample coords -> [B, C, n]."""
    B, C, H, W = x.shape
    h0, w0 = torch.floor(h), torch.floor(w)
    lh, lw = h - h0, w - w0
    h0, w0 = h0.long(), w0.long()
    xf = x.reshape(B, C, H * W)

    def corner(hi, wi):
        ok = (hi >= 0) & (hi <= H - 1) & (wi >= 0) & (wi <= W - 1)
        idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1))[:, None, :].expand(B, C, -1)
        return torch.gather(xf, 2, idx) * ok[:, None, :].to(x.dtype)
    val = ((1 - lh) * (1 - lw))[:, None] * corner(h0, w0) + ((1 - lh) * lw)[:, None] * corner(h0, w0 + 1) \
        + (lh * (1 - lw))[:, None] * corner(h0 + 1, w0) + (lh * lw)[:, None] * corner(h0 + 1, w0 + 1)
    inside = (h > -1) & (w > -1) & (h < H) & (w < W)
    return val * inside[:, None, :].to(x.dtype)


def dcn_v2(x, offset_buf, mask_buf, weight, bias, stride):
    """Modulated deformable 3x3 conv, pad 1, dil 1, groups 1, deformable_groups 1.
    offset_buf: [B, 18, *, *], mask_buf: [B, 9, *, *] -- their spatial dims need NOT equal the output
    dims: like the CUDA kernel (:607-617) tap k of output (h, w) reads flat element
    ((2k)*Ho + h)*Wo + w (dh), ((2k+1)*Ho + h)*Wo + w (dw) and (k*Ho + h)*Wo + w (mask) of the
    per-sample buffer."""
    B, C, H, W = x.shape
    O = weight.shape[0]
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    n = Ho * Wo
    off = offset_buf.reshape(B, -1)
    msk = mask_buf.reshape(B, -1)
    pos = torch.arange(n)
    ho = (pos // Wo).float() * stride - 1
    wo = (pos % Wo).float() * stride - 1
    cols = []
    for k in range(9):
        i, j = divmod(k, 3)
        dh = off[:, (2 * k) * n + pos]
        dw = off[:, (2 * k + 1) * n + pos]
        m = msk[:, k * n + pos]
        cols.append(bilinear_zero(x, ho[None] + i + dh, wo[None] + j + dw) * m[:, None])
    col = torch.stack(cols, 2).reshape(B, C * 9, n)                      # row index = c*9 + (i*3+j)
    out = torch.einsum("ok,bkn->bon", weight.reshape(O, C * 9), col)
    if bias is not None:
        out = out + bias[None, :, None]
    return out.reshape(B, O, Ho, Wo)


# --------------------------------------------------------------------------- DyConv
def h_sigmoid(x):
    return F.relu6(x + 3) / 6


def dyrelu(sd, p, x):
    """layers/dyrelu.py:78-112, K2, use_bias, lambda_a=2, init_a=(1,0), init_b=(0,0)."""
    B, C, _, _ = x.shape
    y = x.mean((2, 3))
    y = h_sigmoid(_lin(sd, p + ".fc.2", F.relu(_lin(sd, p + ".fc.0", y)))).reshape(B, 4 * C, 1, 1)
    a1, b1, a2, b2 = torch.split(y, C, 1)
    a1 = (a1 - 0.5) * 2 + 1.0
    a2 = (a2 - 0.5) * 2
    return torch.max(x * a1 + (b1 - 0.5), x * a2 + (b2 - 0.5))


def dyconv(sd, p, feats, spec):
    """vldyhead.py:205-247.  `p` = '...dyhead_tower.{3i+2}'.  DyConv[0]: from level+1 (stride 1, then
    bilinear up, align_corners=True), DyConv[1]: same level, DyConv[2]: from level-1 (stride 2); all
    three use the offsets/masks computed from the CURRENT level."""
    def conv_gn(k, x, off, msk, stride):
        y = dcn_v2(x, off, msk, sd[f"{p}.DyConv.{k}.conv.weight"], sd[f"{p}.DyConv.{k}.conv.bias"], stride)
        return F.group_norm(y, spec.gn_groups, sd[f"{p}.DyConv.{k}.bn.weight"], sd[f"{p}.DyConv.{k}.bn.bias"],
                            spec.gn_eps)
    out = []
    for lvl, f in enumerate(feats):
        om = F.conv2d(f, sd[p + ".offset.weight"], sd[p + ".offset.bias"], padding=1)
        off, msk = om[:, :18], om[:, 18:].sigmoid()
        branches = [conv_gn(1, f, off, msk, 1)]
        if lvl > 0:
            branches.append(conv_gn(2, feats[lvl - 1], off, msk, 2))
        if lvl < len(feats) - 1:
            up = conv_gn(0, feats[lvl + 1], off, msk, 1)
            branches.append(F.interpolate(up, size=f.shape[-2:], mode="bilinear", align_corners=True))
        res = torch.stack(branches)
        attn = torch.stack([F.relu(F.conv2d(b.mean((2, 3), keepdim=True), sd[p + ".AttnConv.1.weight"],
                                            sd[p + ".AttnConv.1.bias"])) for b in branches])
        out.append((res * h_sigmoid(attn)).mean(0))
    return [dyrelu(sd, p + ".relu", o) for o in out]


# --------------------------------------------------------------------------- head
def vldyhead(sd, p, feats, lang, spec, trace=None):
    """VLDyHead.forward (vldyhead.py:769-900), eval outputs only.  `p` = 'rpn.head'.
    lang: dict with 'hidden' [B,T,768], 'masks' [B,T] int64.
    Returns bbox_reg[5] (B,4,H,W), centerness[5] (B,1,H,W), dot_product_logits[5] (B,HW,T),
    plus the tower outputs (visual feats, text hidden) for layer-wise checks.  trace: optional list that receives, per
    fusion layer, {"fuse_feats", "fuse_hidden", "bert_hidden", "dyconv_feats"} (the parity error ladder)."""
    hidden, masks = lang["hidden"], lang["masks"]
    ext = extended_mask(masks)
    for i in range(spec.dyhead_convs):
        t = f"{p}.dyhead_tower"
        feats, hidden = vl_fuse(sd, f"{t}.{3 * i}.b_attn", feats, hidden, masks, spec)
        rec = {"fuse_feats": feats, "fuse_hidden": hidden}
        hidden = bert_layer(sd, f"{t}.{3 * i + 1}", hidden, ext, spec.bert_heads, spec.bert_eps, clamp=True)
        feats = dyconv(sd, f"{t}.{3 * i + 2}", feats, spec)
        if trace is not None:
            rec.update(bert_hidden=hidden, dyconv_feats=feats)
            trace.append(rec)
    emb = F.normalize(hidden, p=2, dim=-1)
    tok = _lin(sd, p + ".dot_product_projection_text", emb / 2.0)        # [B, T, 256]
    tok_bias = emb @ sd[p + ".bias_lang"] + sd[p + ".bias0"]              # [B, T]
    bbox_reg, ctr, logits = [], [], []
    for l, f in enumerate(feats):
        B, C, H, W = f.shape
        bbox_reg.append(F.conv2d(f, sd[p + ".bbox_pred.weight"], sd[p + ".bbox_pred.bias"])
                        * sd[f"{p}.scales.{l}.scale"])
        ctr.append(F.conv2d(f, sd[p + ".centerness.weight"], sd[p + ".centerness.bias"]))
        q = f.permute(0, 2, 3, 1).reshape(B, H * W, C)
        d = q @ tok.transpose(-1, -2) / math.exp(float(sd[p + ".log_scale"])) + tok_bias[:, None, :]
        logits.append(d.clamp(min=-50000, max=50000))
    return {"bbox_reg": bbox_reg, "centerness": ctr, "dot_product_logits": logits,
            "feats": feats, "hidden": hidden}
