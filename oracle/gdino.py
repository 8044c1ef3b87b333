"""Oracle: multi-scale deformable attention of the MQ-GroundingDINO path (test infrastructure, see oracle/__init__.py).

Restates groundingdino_new/models/GroundingDINO/ms_deform_attn.py:93-133 (`multi_scale_deformable_attn_pytorch`, the
pure-torch statement of the CUDA operator: per level F.grid_sample(bilinear, zeros, align_corners=False) on locations
2*loc - 1, weighted sum over levels x points) and :232-359 (`MultiScaleDeformableAttention.forward`, batch_first: value
projection, key-padding zeroing, sampling offsets and softmax-ed attention weights from the query, sampling locations from
2-d reference points or 4-d reference boxes, output projection).
PINS: tests/golden/msda.npz holds outputs of the reference's own `multi_scale_deformable_attn_pytorch` (executed in place by
oracle/gen_golden.py); on the GPU both this restatement and the HIP kernel are also compared with the reference's CUDA kernel
(ms_deform_im2col_cuda.cuh:237-299 built by oracle/build_ref.py)."""
import torch
import torch.nn.functional as F


def ms_deform_attn_core(value, spatial_shapes, sampling_locations, attention_weights):
    """value [B, S, M, D]; spatial_shapes [(H, W)] * L; loc [B, Q, M, L, P, 2]; attn [B, Q, M, L, P] -> [B, Q, M * D]."""
    B, _, M, D = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in spatial_shapes]
    vals = value.split([h * w for h, w in shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(shapes):
        v = vals[lvl].flatten(2).transpose(1, 2).reshape(B * M, D, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)                          # [B*M, Q, P, 2]
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    a = attention_weights.transpose(1, 2).reshape(B * M, 1, Q, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * a).sum(-1).view(B, M * D, Q)
    return out.transpose(1, 2).contiguous()


def ms_deform_attn(sd, p, query, value, reference_points, spatial_shapes, key_padding_mask=None, query_pos=None,
                   heads=8, levels=4, points=4):
    """MultiScaleDeformableAttention.forward (batch_first).  query [B, Q, C], value [B, S, C] (None -> query),
    reference_points [B, Q, L, 2 or 4] normalised, key_padding_mask [B, S] bool (True = padding)."""
    if value is None:
        value = query
    if query_pos is not None:
        query = query + query_pos
    B, Q, C = query.shape
    S = value.shape[1]
    v = F.linear(value, sd[p + ".value_proj.weight"], sd[p + ".value_proj.bias"])
    if key_padding_mask is not None:
        v = v.masked_fill(key_padding_mask[..., None], 0.0)
    v = v.view(B, S, heads, -1)
    off = F.linear(query, sd[p + ".sampling_offsets.weight"], sd[p + ".sampling_offsets.bias"]).view(B, Q, heads, levels, points, 2)
    aw = F.linear(query, sd[p + ".attention_weights.weight"], sd[p + ".attention_weights.bias"]).view(B, Q, heads, levels * points)
    aw = aw.softmax(-1).view(B, Q, heads, levels, points)
    shp = torch.tensor([[h, w] for h, w in spatial_shapes], dtype=query.dtype)
    if reference_points.shape[-1] == 2:
        norm = torch.stack([shp[:, 1], shp[:, 0]], -1)
        loc = reference_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = reference_points[:, :, None, :, None, :2] + off / points * reference_points[:, :, None, :, None, 2:] * 0.5
    out = ms_deform_attn_core(v, spatial_shapes, loc, aw)
    return F.linear(out, sd[p + ".output_proj.weight"], sd[p + ".output_proj.bias"])


def make_msda_weights(C=256, heads=8, levels=4, points=4, seed=0, prefix="attn"):
    g = torch.Generator().manual_seed(seed)
    n = heads * levels * points
    return {prefix + ".value_proj.weight": torch.randn(C, C, generator=g) / C ** 0.5, prefix + ".value_proj.bias": torch.randn(C, generator=g) * 0.05,
            prefix + ".sampling_offsets.weight": torch.randn(2 * n, C, generator=g) * 0.05, prefix + ".sampling_offsets.bias": torch.randn(2 * n, generator=g) * 2.0,
            prefix + ".attention_weights.weight": torch.randn(n, C, generator=g) * 0.1, prefix + ".attention_weights.bias": torch.randn(n, generator=g) * 0.1,
            prefix + ".output_proj.weight": torch.randn(C, C, generator=g) / C ** 0.5, prefix + ".output_proj.bias": torch.randn(C, generator=g) * 0.05}


# ================================================================================================ MQ-GroundingDINO model
# Restates groundingdino_new/models/GroundingDINO/{groundingdino.py:438-661,291-335; transformer.py:211-400,467-594,644-736,
# 764-927; fuse_modules.py:99-297; transformer_vanilla.py:65-123; utils.py:18-108,178-268; bertwarper.py:273-320;
# backbone/position_encoding.py:76-125; backbone/backbone.py:135-146; util/misc.py:474-487,721-725}.
# PINS: tests/golden/gdino_{vq,text,convert}.npz = outputs of the reference's own GroundingDINO module built by
# oracle/gen_golden_gdino.py (every stage below is compared in tests/test_oracle_golden.py).
import math

from . import backbone as obackbone
from . import language as olang
from .detector import labels_and_maps, pooled_fpn_tokens, select_queries


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def _mlp(sd, p, x, n):
    """utils.py MLP: Linear + ReLU ... Linear."""
    for i in range(n):
        x = _lin(sd, f"{p}.layers.{i}", x)
        if i < n - 1:
            x = F.relu(x)
    return x


def inverse_sigmoid(x, eps=1e-3):
    """util/misc.py:721-725."""
    x = x.clamp(0, 1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def nested_masks(image_sizes, H, W):
    """util/misc.py:474-487: padding mask [B,H,W] (True = padding) from the (h, w) of every image."""
    m = torch.zeros(len(image_sizes), H, W, dtype=torch.bool)
    for i, (h, w) in enumerate(image_sizes):
        m[i, int(h):, :] = True
        m[i, :, int(w):] = True
    return m


def resize_mask(mask, size):
    """F.interpolate(mask[None].float(), size).to(bool)[0] (nearest), swin_transformer.py:743-747, groundingdino.py:490-491."""
    return F.interpolate(mask[None].float(), size=tuple(size)).to(torch.bool)[0]


def position_embedding(mask, num_pos_feats, temperature):
    """PositionEmbeddingSineHW (normalize=True, scale 2 pi), position_encoding.py:76-125 -> [B, 2*num_pos_feats, H, W]."""
    nm = ~mask
    y = nm.cumsum(1, dtype=torch.float32)
    x = nm.cumsum(2, dtype=torch.float32)
    y = y / (y[:, -1:, :] + 1e-6) * (2 * math.pi)
    x = x / (x[:, :, -1:] + 1e-6) * (2 * math.pi)
    d = torch.arange(num_pos_feats, dtype=torch.float32)
    d = temperature ** (2 * torch.div(d, 2, rounding_mode="floor") / num_pos_feats)
    px, py = x[..., None] / d, y[..., None] / d
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), 4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), 4).flatten(3)
    return torch.cat((py, px), 3).permute(0, 3, 1, 2)


def sine_pos_embed(pos, num_pos_feats=128, temperature=10000, exchange_xy=True):
    """utils.py get_sine_pos_embed: [..., n] -> [..., n*num_pos_feats]."""
    d = torch.arange(num_pos_feats, dtype=torch.float32)
    d = temperature ** (2 * torch.div(d, 2, rounding_mode="floor") / num_pos_feats)
    res = []
    for x in pos.split([1] * pos.shape[-1], -1):
        s = x * (2 * math.pi) / d
        res.append(torch.stack((s[..., 0::2].sin(), s[..., 1::2].cos()), 3).flatten(2))
    if exchange_xy:
        res[0], res[1] = res[1], res[0]
    return torch.cat(res, -1)


def sine_embed_for_boxes(box):
    """utils.py gen_sineembed_for_position for 4-d [nq, bs, 4] (cx, cy, w, h) -> [nq, bs, 512] in (y, x, w, h) order."""
    d = torch.arange(128, dtype=torch.float32)
    d = 10000 ** (2 * torch.div(d, 2, rounding_mode="floor") / 128)

    def emb(v):
        s = v[:, :, None] * (2 * math.pi) / d
        return torch.stack((s[:, :, 0::2].sin(), s[:, :, 1::2].cos()), 3).flatten(2)
    return torch.cat((emb(box[:, :, 1]), emb(box[:, :, 0]), emb(box[:, :, 2]), emb(box[:, :, 3])), 2)


def special_token_masks(input_ids, special_ids):
    """bertwarper.py:273-320 generate_masks_with_special_tokens_and_transfer_map: block-diagonal self-attention mask [B,T,T]
    (True = may attend) between consecutive special tokens ([CLS], [SEP], '.', '?') and positions restarting in every block."""
    B, T = input_ids.shape
    special = torch.zeros(B, T, dtype=torch.bool)
    for s in special_ids:
        special |= input_ids == s
    attn = torch.eye(T, dtype=torch.bool)[None].repeat(B, 1, 1)
    pos = torch.zeros(B, T, dtype=torch.long)
    prev = 0                                                   # NOT reset per row, like the reference
    for row, col in torch.nonzero(special).tolist():
        if col == 0 or col == T - 1:
            attn[row, col, col] = True
            pos[row, col] = 0
        else:
            attn[row, prev + 1:col + 1, prev + 1:col + 1] = True
            pos[row, prev + 1:col + 1] = torch.arange(0, col - prev)
        prev = col
    return attn, pos


def multihead_attention(sd, p, q, k, v, heads, attn_mask=None, key_padding_mask=None):
    """torch.nn.MultiheadAttention forward (batch-first here: q [B,Nq,C], k/v [B,Nk,C]); attn_mask bool [B*heads or 1, Nq, Nk]
    with True = NOT allowed (indexed b*heads + h, torch's convention), key_padding_mask bool [B,Nk] True = ignore."""
    B, Nq, C = q.shape
    Nk, hd = k.shape[1], C // heads
    W, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    qh = F.linear(q, W[:C], b[:C]).view(B, Nq, heads, hd).transpose(1, 2)
    kh = F.linear(k, W[C:2 * C], b[C:2 * C]).view(B, Nk, heads, hd).transpose(1, 2)
    vh = F.linear(v, W[2 * C:], b[2 * C:]).view(B, Nk, heads, hd).transpose(1, 2)
    s = (qh / math.sqrt(hd)) @ kh.transpose(-1, -2)
    if attn_mask is not None:
        m = attn_mask.view(-1, heads, Nq, Nk) if attn_mask.shape[0] == B * heads else attn_mask[None]
        s = s.masked_fill(m, float("-inf"))
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    o = (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, Nq, C)
    return _lin(sd, p + ".out_proj", o)


def bi_attention_block(sd, p, v, l, mask_v, mask_l, heads):
    """fuse_modules.py:99-297 BiAttentionBlock / BiMultiHeadAttention (eval): pre-LN both sides, image->text and text->image
    attention from ONE logit tensor, +-50000 clamps, layer scale gamma.  mask_v [B,N], mask_l [B,T] bool True = padding."""
    v = _ln(sd, p + ".layer_norm_v", v)
    l = _ln(sd, p + ".layer_norm_l", l)
    a = p + ".attn"
    B, N, _ = v.shape
    E = sd[a + ".v_proj.weight"].shape[0]
    hd = E // heads

    def shape(x):
        return x.view(B, -1, heads, hd).transpose(1, 2)
    q = shape(_lin(sd, a + ".v_proj", v) * hd ** -0.5)
    k = shape(_lin(sd, a + ".l_proj", l))
    val_v = shape(_lin(sd, a + ".values_v_proj", v))
    val_l = shape(_lin(sd, a + ".values_l_proj", l))
    w = q @ k.transpose(-1, -2)                                            # [B, h, N, T]
    w = (w - w.max()).clamp(min=-50000, max=50000)
    wl = w.transpose(-1, -2)
    wl = (wl - wl.max(-1, keepdim=True)[0]).clamp(min=-50000, max=50000)
    if mask_v is not None:
        wl = wl.masked_fill(mask_v[:, None, None, :], float("-inf"))
    wl = wl.softmax(-1)
    if mask_l is not None:
        w = w.masked_fill(mask_l[:, None, None, :], float("-inf"))
    wv = w.softmax(-1)
    dv = _lin(sd, a + ".out_v_proj", (wv @ val_l).transpose(1, 2).reshape(B, N, E))
    dl = _lin(sd, a + ".out_l_proj", (wl @ val_v).transpose(1, 2).reshape(B, -1, E))
    return v + sd[p + ".gamma_v"] * dv, l + sd[p + ".gamma_l"] * dl


def text_enhancer_layer(sd, p, x, self_masks, pos, heads):
    """transformer_vanilla.py:92-123 TransformerEncoderLayer (post-norm, ReLU): q = k = x + pos, v = x, attn_mask =
    ~text_self_attention_masks REPEATED `nhead` times along dim 0 (:108-109) and then read by nn.MultiheadAttention as
    [b*nhead + h] -- for B > 1 head (b, h) therefore uses the mask of batch element (b*nhead + h) % B (reference quirk; the
    same for every element when all captions are equal, as in the evaluation loop)."""
    B = x.shape[0]
    m = (~self_masks).repeat(heads, 1, 1) if B > 0 else None
    qk = x + pos
    x = _ln(sd, p + ".norm1", x + multihead_attention(sd, p + ".self_attn", qk, qk, x, heads, attn_mask=m))
    h = _lin(sd, p + ".linear2", F.relu(_lin(sd, p + ".linear1", x)))
    return _ln(sd, p + ".norm2", x + h)


def encoder_reference_points(shapes, valid_ratios):
    """transformer.py:467-481 -> [B, sum HW, L, 2]."""
    refs = []
    for lvl, (H, W) in enumerate(shapes):
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
        ry = ry.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
        rx = rx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
        refs.append(torch.stack((rx, ry), -1))
    return torch.cat(refs, 1)[:, :, None] * valid_ratios[:, None]


def deformable_encoder_layer(sd, p, src, pos, ref, shapes, mask, spec):
    """transformer.py:739-804 (post-norm): MSDeformAttn(query = src + pos, value = src), FFN ReLU."""
    a = ms_deform_attn(sd, p + ".self_attn", src + pos, src, ref, shapes, mask, None, spec.nheads, spec.levels, spec.points)
    src = _ln(sd, p + ".norm1", src + a)
    h = _lin(sd, p + ".linear2", F.relu(_lin(sd, p + ".linear1", src)))
    return _ln(sd, p + ".norm2", src + h)


def encoder_output_proposals(memory, mask, shapes):
    """utils.py:52-108 gen_encoder_output_proposals (learnedwh None) -> (output_memory, output_proposals unsigmoided)."""
    B = memory.shape[0]
    props, cur = [], 0
    for lvl, (H, W) in enumerate(shapes):
        m = mask[:, cur:cur + H * W].view(B, H, W, 1)
        vh = (~m[:, :, 0, 0]).sum(1)
        vw = (~m[:, 0, :, 0]).sum(1)
        gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
        grid = torch.cat([gx[..., None], gy[..., None]], -1)
        scale = torch.cat([vw[:, None], vh[:, None]], 1).view(B, 1, 1, 2)
        grid = (grid[None].expand(B, -1, -1, -1) + 0.5) / scale
        wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
        props.append(torch.cat((grid, wh), -1).view(B, -1, 4))
        cur += H * W
    props = torch.cat(props, 1)
    valid = ((props > 0.01) & (props < 0.99)).all(-1, keepdim=True)
    props = torch.log(props / (1 - props))
    props = props.masked_fill(mask[..., None], float("inf")).masked_fill(~valid, float("inf"))
    mem = memory.masked_fill(mask[..., None], 0.0).masked_fill(~valid, 0.0)
    return mem, props


def contrastive_embed(x, text, text_token_mask, max_text_len):
    """utils.py:241-268 ContrastiveEmbed: x . text^T, padding tokens -inf, padded to max_text_len with -inf."""
    res = x @ text.transpose(-1, -2)
    res = res.masked_fill(~text_token_mask[:, None, :], float("-inf"))
    out = torch.full((*res.shape[:-1], max_text_len), float("-inf"))
    out[..., :res.shape[-1]] = res
    return out


def decoder_layer(sd, p, tgt, query_pos, ref_input, memory_text, text_pad_mask, memory, mem_mask, shapes, spec):
    """transformer.py:807-927 (batch-first here): self-attention, text cross-attention, deformable cross-attention, FFN."""
    qk = tgt + query_pos
    tgt = _ln(sd, p + ".norm2", tgt + multihead_attention(sd, p + ".self_attn", qk, qk, tgt, spec.nheads))
    t2 = multihead_attention(sd, p + ".ca_text", tgt + query_pos, memory_text, memory_text, spec.nheads, key_padding_mask=text_pad_mask)
    tgt = _ln(sd, p + ".catext_norm", tgt + t2)
    t2 = ms_deform_attn(sd, p + ".cross_attn", tgt + query_pos, memory, ref_input, shapes, mem_mask, None, spec.nheads, spec.levels,
                        spec.points)
    tgt = _ln(sd, p + ".norm1", tgt + t2)
    h = _lin(sd, p + ".linear2", F.relu(_lin(sd, p + ".linear1", tgt)))
    return _ln(sd, p + ".norm3", tgt + h)


def transformer(sd, spec, srcs, masks, poss, text, topk_override=None):
    """Transformer.forward (transformer.py:211-400), two_stage_type 'standard', embed_init_tgt.  srcs / poss: per level
    [B,C,H,W]; masks: per level [B,H,W]; text: dict(encoded_text [B,T,C], text_token_mask [B,T] bool True = real token,
    position_ids [B,T], text_self_attention_masks [B,T,T]).  Returns a dict of every stage.
    topk_override [B, nq] (tests only): decode THESE proposals instead of the oracle's own top-k (kept as `topk_own`) -- the
    selection is discontinuous and rank-dependent (query slot r gets tgt_embed[r]), so a device implementation is compared
    downstream of it on identical selections, and on the selection itself as a set."""
    t = "transformer"
    shapes = [tuple(s.shape[-2:]) for s in srcs]
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    mask = torch.cat([m.flatten(1) for m in masks], 1)
    pos = torch.cat([p_.flatten(2).transpose(1, 2) + sd[t + ".level_embed"][l].view(1, 1, -1) for l, p_ in enumerate(poss)], 1)
    B = src.shape[0]

    def valid_ratio(m):
        H, W = m.shape[1:]
        return torch.stack([(~m[:, 0, :]).sum(1).float() / W, (~m[:, :, 0]).sum(1).float() / H], -1)
    vr = torch.stack([valid_ratio(m) for m in masks], 1)                              # [B, L, 2]
    # ---- encoder (transformer.py:482-594): fusion -> text enhancer -> deformable layer, per layer
    mem, mtext = src, text["encoded_text"]
    text_pad = ~text["text_token_mask"]
    ref = encoder_reference_points(shapes, vr)
    pos_text = sine_pos_embed(text["position_ids"][..., None].float(), num_pos_feats=256, exchange_xy=False)
    for i in range(spec.enc_layers):
        mem, mtext = bi_attention_block(sd, f"{t}.encoder.fusion_layers.{i}", mem, mtext, mask, text_pad, spec.nheads // 2)
        mtext = text_enhancer_layer(sd, f"{t}.encoder.text_layers.{i}", mtext, text["text_self_attention_masks"], pos_text,
                                    spec.nheads // 2)
        mem = deformable_encoder_layer(sd, f"{t}.encoder.layers.{i}", mem, pos, ref, shapes, mask, spec)
    out = {"memory": mem, "memory_text": mtext, "mask": mask, "shapes": shapes}
    # ---- two-stage query selection (transformer.py:262-306)
    omem, props = encoder_output_proposals(mem, mask, shapes)
    omem = _ln(sd, t + ".enc_output_norm", _lin(sd, t + ".enc_output", omem))
    cls = contrastive_embed(omem, mtext, text["text_token_mask"], spec.max_text_len)
    coord = _mlp(sd, t + ".enc_out_bbox_embed", omem, 3) + props
    topk = torch.topk(cls.max(-1)[0], spec.num_queries, dim=1)[1]
    out["topk_own"], out["topk_logits"] = topk, cls.max(-1)[0]
    if topk_override is not None:
        topk = topk_override
    refpoint = torch.gather(coord, 1, topk[..., None].repeat(1, 1, 4))
    out["topk"] = topk
    out["init_box"] = torch.gather(props, 1, topk[..., None].repeat(1, 1, 4)).sigmoid()
    out["hs_enc"] = torch.gather(omem, 1, topk[..., None].repeat(1, 1, spec.hidden))
    out["ref_enc"] = refpoint.sigmoid()
    tgt = sd[t + ".tgt_embed.weight"][None].repeat(B, 1, 1)
    # ---- decoder (transformer.py:644-736), batch-first
    refs = [refpoint.sigmoid()]
    hs = []
    rp = refs[0]
    for i in range(spec.dec_layers):
        ref_in = rp[:, :, None] * torch.cat([vr, vr], -1)[:, None]                     # [B, nq, L, 4]
        sine = sine_embed_for_boxes(ref_in[:, :, 0, :].transpose(0, 1)).transpose(0, 1)
        qpos = _mlp(sd, t + ".decoder.ref_point_head", sine, 2)
        tgt = decoder_layer(sd, f"{t}.decoder.layers.{i}", tgt, qpos, ref_in, mtext, text_pad, mem, mask, shapes, spec)
        rp = (_mlp(sd, f"{t}.decoder.bbox_embed.{i}", tgt, 3) + inverse_sigmoid(rp)).sigmoid()
        refs.append(rp)
        hs.append(_ln(sd, t + ".decoder.norm", tgt))
    out["hs"], out["refs"] = hs, refs
    return out


def convert_to_glip_output(prob, boxes, positive_map, image_sizes, num_classes, box_threshold):
    """GroundingDINO.convert_groundingdino_to_glip_output (groundingdino.py:291-335) with convert_grounding_to_od_logits
    (rpn/inference.py:772-790, MEAN), BoxList.clip_to_image(remove_empty=False) and remove_small_boxes(min_size=0).
    prob [B,N,T] sigmoided token scores, boxes [B,N,4] cxcywh in 0..1; image_sizes [(h, w)].
    Returns per image (boxes xyxy [n,4], scores [n], labels [n]).  A label whose token list is empty yields NaN class scores
    (mean over nothing), the per-query max over classes is then NaN for EVERY query and nothing passes the threshold."""
    B, N, _ = prob.shape
    scores = torch.zeros(B, N, num_classes - 1)
    for lab, toks in positive_map.items():
        scores[:, :, lab - 1] = prob[:, :, torch.LongTensor(list(toks))].mean(-1)
    res = []
    for b in range(B):
        keep = scores[b].max(-1)[0] > box_threshold
        sc, idx = scores[b][keep].max(-1) if keep.any() else (scores.new_zeros(0), torch.zeros(0, dtype=torch.long))
        H, W = image_sizes[b]
        bx = boxes[b][keep].view(-1, 4) * torch.tensor([W, H, W, H], dtype=torch.float32)
        xy1 = bx[:, :2] - bx[:, 2:] / 2
        xy2 = bx[:, 2:] + xy1
        out = torch.cat([xy1, xy2], 1)
        out[:, 0::2] = out[:, 0::2].clamp(0, W - 1)
        out[:, 1::2] = out[:, 1::2].clamp(0, H - 1)
        ok = ((out[:, 2] - out[:, 0] + 1) >= 0) & ((out[:, 3] - out[:, 1] + 1) >= 0)
        res.append((out[ok], sc[ok], idx[ok] + 1))
    return res


def forward(sd, spec, images, image_sizes, input_ids, attention_mask, positive_map, special_ids, bank=None, topk_override=None):
    """GroundingDINO.forward, eval (groundingdino.py:438-623).  images [B,3,H,W] padded (ImageList.tensors), image_sizes
    [(h, w)], input_ids / attention_mask [B, Ttok] = the tokenizer's output with padding='max_length' (cut to max_text_len
    here like :530-537), special_ids = ids of [CLS], [SEP], '.', '?' (:194).  Returns a dict with every stage and `detections`."""
    B, _, H, W = images.shape
    full = nested_masks(image_sizes, H, W)
    feats = obackbone.swin_forward(sd, "backbone.0", images, spec)[1:]
    masks = [resize_mask(full, f.shape[-2:]) for f in feats]
    srcs = []
    for l in range(spec.levels):
        if l < len(feats):
            x = F.conv2d(feats[l], sd[f"input_proj.{l}.0.weight"], sd[f"input_proj.{l}.0.bias"])
        else:
            x = F.conv2d(feats[-1] if l == len(feats) else srcs[-1], sd[f"input_proj.{l}.0.weight"], sd[f"input_proj.{l}.0.bias"],
                         stride=2, padding=1)
            masks.append(resize_mask(full, x.shape[-2:]))
        srcs.append(F.group_norm(x, spec.gn_groups, sd[f"input_proj.{l}.1.weight"], sd[f"input_proj.{l}.1.bias"], 1e-5))
    poss = [position_embedding(m, spec.hidden // 2, spec.pe_temperature) for m in masks]
    out = {"swin": feats, "masks": masks, "srcs": srcs, "pos": poss}
    # ---- text (groundingdino.py:496-571)
    self_masks, position_ids = special_token_masks(input_ids, special_ids)
    T = spec.max_text_len
    self_masks, position_ids = self_masks[:, :T, :T], position_ids[:, :T]
    ids, tok_mask = input_ids[:, :T], attention_mask[:, :T].bool()
    vision = vmask = image_tokens = None
    if spec.vision_query and bank is not None:
        assert B == 1, "groundingdino.py:502: vision queries only for batch size 1"
        labels, all_map = labels_and_maps(positive_map, T)
        vision, vmask = select_queries(bank, [labels], [all_map], spec.num_query_per_class)
        image_tokens = pooled_fpn_tokens(srcs)
    hidden = olang.qv_bert(sd, "bert", ids, self_masks.float(), vision, image_tokens, vmask, spec, position_ids=position_ids)
    out["bert"] = hidden[-1]
    text = {"encoded_text": _lin(sd, "feat_map", hidden[-1]), "text_token_mask": tok_mask, "position_ids": position_ids,
            "text_self_attention_masks": self_masks}
    out["encoded_text"] = text["encoded_text"]
    out.update(transformer(sd, spec, srcs, masks, poss, text, topk_override))
    # ---- heads of the last decoder layer (groundingdino.py:585-604,641-642)
    hs, refs = out["hs"], out["refs"]
    out["pred_boxes"] = (_mlp(sd, f"bbox_embed.{spec.dec_layers - 1}", hs[-1], 3) + inverse_sigmoid(refs[-2])).sigmoid()
    out["pred_logits"] = contrastive_embed(hs[-1], out["memory_text"], tok_mask, T).sigmoid()
    out["detections"] = convert_to_glip_output(out["pred_logits"], out["pred_boxes"], positive_map, image_sizes, spec.num_classes,
                                               spec.box_threshold)
    return out
