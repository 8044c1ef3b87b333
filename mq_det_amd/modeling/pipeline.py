"""Functional MI355X forward of MQ-GLIP over an fp16 "plan" (packed inference weights).

Activations that feed MFMA contractions are fp16 (fp32 accumulation inside every kernel / GEMM); the RESIDUAL STREAMS of the
transformer stacks (Swin tokens, BERT / GCP / VLDyHead text hidden states) are fp32 (MODEL.RESIDUAL_FP32, default on); feature
maps are NHWC in memory.
Hand-written HIP kernels (mq_det_amd.ops -> libmqdet_hip.so) do: Swin window attention, every dense
attention (BERT, GCP pre-select, the two VLFuse directions), the GCP sparse cross-attention + gated residual,
LayerNorm (+ fused residual add), every 3x3 convolution, DCNv2 (gather + blend + MFMA + GroupNorm statistics in
one kernel), the DyConv / DyReLU epilogue, alignment scoring, box decode and class-aware NMS.  Library GEMMs
(hipBLASLt through torch) do the plain projections; no MIOpen call is on the path.  Independent work is forked
onto side HIP streams (pyramid levels, text chain).  Reference call stack: SURVEY.md section 3.3; each function
cites the reference lines it re-implements.  Nothing here imports the oracle, and nothing runs on CPU.
"""
import math
import os

import torch
import torch.nn.functional as F

from .. import ops

NEG = -1.0e30


# ----------------------------------------------------------------------------- plan
def _pack_swin(P, sd, SW, p, device, dtype):
    """Swin inference tensors under prefix `p`: the relative-position bias gathered once (the reference re-gathers it every
    call, swint.py:124-126) and padded to the kernel's window length; fc2 weights with permuted k-slots for the fused MLP."""
    def h(name):
        return sd[name].detach().to(device=device, dtype=dtype).contiguous()
    ws = SW.WINDOW_SIZE
    N = ws * ws
    for i, (depth, heads) in enumerate(zip(SW.DEPTHS, SW.NUM_HEADS)):
        for j in range(depth):
            b = f"{p}.layers.{i}.blocks.{j}.attn"
            idx = sd[b + ".relative_position_index"].to(device).reshape(-1)
            rel = sd[b + ".relative_position_bias_table"].detach().to(device=device, dtype=torch.float32)[idx]
            rel = rel.reshape(N, N, heads).permute(2, 0, 1)
            NP = ops.window_pad(ws)                                                        # 64 (window 7) or 160 (window 12)
            P[b + ".rel_bias"] = F.pad(rel, (0, NP - N, 0, NP - N)).contiguous()         # [heads, NP, NP], see ops.pad_rel_bias
    # Swin MLP halves that run as one fused kernel (mq_swin_mlp2_fwd): both weights fragment-major.  KERNELS["SWIN_MLP_VARIANT"] = 1 selects the
    # library path (LayerNorm kernel + two GEMMs + GELU); the first-generation kernel mq_swin_mlp_fwd was removed in round 5 (lost its A/Bs)
    P["_swin_fused_mlp"] = bool(SW.get("FUSED_MLP", True)) and P["_r32"] and ops.KERNELS["SWIN_MLP_VARIANT"] == 2
    P["_swin_fused_widths"] = tuple(w for w in SW.get("FUSED_MLP_WIDTHS", ops.SWIN_MLP_WIDTHS) if w in ops.SWIN_MLP_WIDTHS)
    for i, depth in enumerate(SW.DEPTHS):
        Ci = SW.EMBED_DIM * 2 ** i
        if P["_swin_fused_mlp"] and Ci in P["_swin_fused_widths"]:
            for j in range(depth):
                b = f"{p}.layers.{i}.blocks.{j}.mlp"
                P[b + ".w1f"], P[b + ".w2f"] = ops.swin_mlp2_pack(h(b + ".fc1.weight"), h(b + ".fc2.weight"))
    # patch embedding (4x4 stride-4 conv) as a GEMM over (kh, kw, c)-ordered patches
    w = sd[p + ".patch_embed.proj.weight"].detach().to(device=device, dtype=torch.float32)
    P[p + ".patch_embed.lin"] = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(dtype).contiguous()
    if w.shape[0] in (96, 192) and tuple(w.shape[1:]) == (3, 4, 4):          # operands of mq_patch_embed_fwd (projection + both LayerNorms)
        P[p + ".patch_embed.wpk"] = ops.patch_embed_pack(w).to(dtype).contiguous()
        P[p + ".patch_embed.wpk_nchw"] = ops.patch_embed_pack(w, nchw=True).to(dtype).contiguous()     # for fp32 NCHW pixels (the caller's tensor)
        P[p + ".patch_embed.f32"] = [sd[n].detach().to(device=device, dtype=torch.float32).contiguous() for n in
                                     (p + ".patch_embed.proj.bias", p + ".patch_embed.norm.weight", p + ".patch_embed.norm.bias",
                                      p + ".layers.0.blocks.0.norm1.weight", p + ".layers.0.blocks.0.norm1.bias")]


def _pack_language(P, sd, cfg, p, device, dtype):
    """BERT + GCP inference tensors under prefix `p`: fused q|k projections, gates folded, pre-select k / v split."""
    def h(name):
        return sd[name].detach().to(device=device, dtype=dtype).contiguous()

    def f32(name):
        return sd[name].detach().to(device=device, dtype=torch.float32).contiguous()
    LB = cfg.MODEL.LANGUAGE_BACKBONE
    nl = LB.get("NUM_HIDDEN_LAYERS", 12)
    for i in range(nl):
        _pack_bert_layer(P, sd, f"{p}.encoder.layer.{i}", device, dtype)
    if cfg.VISION_QUERY.ENABLED:
        qv0 = LB.get("QV_START", 6)
        for i in range(nl - qv0):
            b = f"{p}.encoder.qv_layer.{i}"
            # tanh(ff_gate) folded into the last FFN projection (modeling_bert_new.py:373)
            P[b + ".ff.linear2.gated"] = (f32(b + ".ff.linear2.weight") * torch.tanh(f32(b + ".ff_gate"))).to(dtype)
            P[b + ".attn_gate.w2"] = h(b + ".attn_gate.linear2.weight").reshape(-1)
            # the three weights mq_gcp_attn_fwd streams, in MFMA B-fragment order (one load instruction = 1 KiB of consecutive bytes)
            for n in (".attn.to_q", ".attn.to_out", ".attn_gate.linear1"):
                _pack_frag(P, b + n + ".frag", h(b + n + ".weight"))
        for i in range(2):
            b = f"{p}.pre_select.layers.{i}.image_condition"
            wkv = h(b + ".to_kv.weight")
            half = wkv.shape[0] // 2
            P[b + ".to_k.weight"], P[b + ".to_v.weight"] = wkv[:half].contiguous(), wkv[half:].contiguous()


def _pack_frag(P, key, w):
    if w.shape[0] % 16 == 0 and w.shape[1] % 32 == 0:
        P[key] = ops.pack_b_fragments(w)


def _pack_bert_layer(P, sd, b, device, dtype):
    def h(name):
        return sd[name].detach().to(device=device, dtype=dtype).contiguous()
    P[b + ".qk.weight"] = torch.cat([h(b + ".attention.self.query.weight"), h(b + ".attention.self.key.weight")], 0)
    P[b + ".qk.bias"] = torch.cat([h(b + ".attention.self.query.bias"), h(b + ".attention.self.key.bias")], 0)
    # one projection for q | k | v (KERNELS["BERT_QKV_FUSED"]: mq_attn_text_fwd reads V row-major, no V^T operand)
    P[b + ".qkv.weight"] = torch.cat([P[b + ".qk.weight"], h(b + ".attention.self.value.weight")], 0)
    P[b + ".qkv.bias"] = torch.cat([P[b + ".qk.bias"], h(b + ".attention.self.value.bias")], 0)
    _pack_frag(P, b + ".qkv.frag", P[b + ".qkv.weight"])      # the same matrix in MFMA B-fragment order: what mq_bert_attn_qkv_fwd streams


def _dcn_tiles(P, key):
    """DCNv2 / FPN conv weights of the plan a SECOND time in LDS-tile order (`key.tiles`) when the kernel selection streams them by LDS-DMA
    (KERNELS["DCN_BDMA"]); `key.packed` stays row-major for mq_conv3x3_fwd and for a selection without the switch."""
    if ops.dcn_bdma():
        P[key + ".tiles"] = ops.dcn_weight_tiles(P[key + ".packed"])


def _dcn_w(P, key):
    """The DCNv2 weight operand of `key` under the ACTIVE selection."""
    return P[key + ".tiles"] if (ops.dcn_bdma() and (key + ".tiles") in P) else P[key + ".packed"]


def build_plan(sd, cfg, device, dtype=torch.float16):
    """Pack the fp32 state_dict into inference tensors: fp16 casts, fused / folded / re-laid-out weights."""
    P = {}
    M = cfg.MODEL
    P["_r32"] = bool(M.get("RESIDUAL_FP32", True))        # fp32 residual streams (Swin tokens, text hidden states)

    def h(name):
        return sd[name].detach().to(device=device, dtype=dtype).contiguous()

    def f32(name):
        return sd[name].detach().to(device=device, dtype=torch.float32).contiguous()
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            P[k] = v.detach().to(device=device, dtype=dtype).contiguous()
    _pack_swin(P, sd, M.SWINT, "backbone.body", device, dtype)
    # convs: channels_last weights
    for k in list(P):
        if torch.is_tensor(P[k]) and P[k].dim() == 4:
            P[k] = P[k].contiguous(memory_format=torch.channels_last)
    _pack_language(P, sd, cfg, "language_backbone.body.model", device, dtype)

    def bert(b):
        _pack_bert_layer(P, sd, b, device, dtype)
    # VLDyHead
    D = M.DYHEAD
    hd = 2048 // 8
    for i in range(D.NUM_CONVS):
        b = f"rpn.head.dyhead_tower.{3 * i}.b_attn"
        sc = hd ** -0.5
        P[b + ".q.weight"] = (f32(b + ".attn.v_proj.weight") * sc).to(dtype)        # scale folded (fuse_helper.py:221)
        P[b + ".q.bias"] = (f32(b + ".attn.v_proj.bias") * sc).to(dtype)
        gv, gl = f32(b + ".gamma_v"), f32(b + ".gamma_l")                            # layer scale folded (:424-425)
        P[b + ".ov.weight"] = (f32(b + ".attn.out_v_proj.weight") * gv[:, None]).to(dtype)
        P[b + ".ov.bias"] = (f32(b + ".attn.out_v_proj.bias") * gv).to(dtype)
        P[b + ".ol.weight"] = (f32(b + ".attn.out_l_proj.weight") * gl[:, None]).to(dtype)
        P[b + ".ol.bias"] = (f32(b + ".attn.out_l_proj.bias") * gl).to(dtype)
        # folded projections (DESIGN.md "VLFuse folding"): the image tokens are never projected to 2048-d.
        #   logits_h = LN(v) . (Wq_h^T k_h)          -> Wq8 [8, 256 hd, 256 in] folds into the (tiny) text keys
        #   text side: sum_i P_h[t,i] (Wvv_h LN(v)_i + b) = Wvv_h (P_h^T LN(v))[t] + b, then out_l_proj and
        #   the layer scale -> ONE static [768, 8*256] weight applied to the per-head pooled image features
        #   image side: sum_h P_h (val_l_h) Wov_h^T = sum_h P_h (val_l_h Wov_h^T): out_v_proj * gamma_v folds into the text
        #   values -> Wov8 [8, 256 hd, 256 out]; the kernel's per-head outputs are then just summed over heads
        P[b + ".Wov8"] = P[b + ".ov.weight"].reshape(-1, 8, hd).permute(1, 2, 0).contiguous()
        P[b + ".Wq8"] = P[b + ".q.weight"].reshape(8, hd, -1).contiguous()
        P[b + ".bq8"] = (f32(b + ".attn.v_proj.bias") * sc).reshape(8, hd).contiguous()
        wol = (f32(b + ".attn.out_l_proj.weight") * gl[:, None]).reshape(-1, 8, hd)            # [768, h, e]
        wvv = f32(b + ".attn.values_v_proj.weight").reshape(8, hd, -1)                          # [h, e, c]
        P[b + ".olc.weight"] = torch.einsum("ohe,hec->ohc", wol, wvv).reshape(wol.shape[0], -1).to(dtype).contiguous()
        P[b + ".olc.bias"] = (torch.einsum("ohe,he->o", wol, f32(b + ".attn.values_v_proj.bias").reshape(8, hd))
                              + f32(b + ".attn.out_l_proj.bias") * gl).to(dtype)
        # text operands of the layer as ONE projection of LN(l): folded keys Kf_h = (l Wl_h^T + bl_h) Wq8_h, folded values
        # Vo_h = (l Wvl_h^T + bvl_h) Wov8_h and the per-(head, key) logit bias (l Wl_h^T + bl_h) . bq8_h  (fp32 folding, one
        # fp16 rounding): [2048 | 2048 | 8 (+ 8 zero rows)] x 768
        wl = f32(b + ".attn.l_proj.weight").reshape(8, hd, -1)                                      # [h, e, 768]
        bl = f32(b + ".attn.l_proj.bias").reshape(8, hd)
        wvl = f32(b + ".attn.values_l_proj.weight").reshape(8, hd, -1)
        bvl = f32(b + ".attn.values_l_proj.bias").reshape(8, hd)
        wq8 = (f32(b + ".attn.v_proj.weight") * sc).reshape(8, hd, -1)                              # [h, e, c_in]
        bq8 = (f32(b + ".attn.v_proj.bias") * sc).reshape(8, hd)
        wov8 = (f32(b + ".attn.out_v_proj.weight") * gv[:, None]).reshape(-1, 8, hd).permute(1, 2, 0)   # [h, e, c_out]
        w_kf = torch.einsum("hec,hek->hck", wq8, wl).reshape(8 * wq8.shape[2], -1)                 # [h*c_in, 768]
        b_kf = torch.einsum("hec,he->hc", wq8, bl).reshape(-1)
        w_vo = torch.einsum("heo,hek->hok", wov8, wvl).reshape(8 * wov8.shape[2], -1)
        b_vo = torch.einsum("heo,he->ho", wov8, bvl).reshape(-1)
        w_b = torch.einsum("he,hek->hk", bq8, wl)                                                    # [8, 768]
        b_b = (bq8 * bl).sum(-1)
        z = w_b.new_zeros(8, w_b.shape[1])
        P[b + ".tprep.weight"] = torch.cat([w_kf, w_vo, w_b, z], 0).to(dtype).contiguous()
        P[b + ".tprep.bias"] = torch.cat([b_kf, b_vo, b_b, b_b.new_zeros(8)], 0).to(dtype).contiguous()
        bert(f"rpn.head.dyhead_tower.{3 * i + 1}")
        b = f"rpn.head.dyhead_tower.{3 * i + 2}"
        for k in range(3):
            w = f32(f"{b}.DyConv.{k}.conv.weight")                                   # [O, C, 3, 3] -> [O, tap*C + c]
            P[f"{b}.DyConv.{k}.packed"] = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(dtype).contiguous()
            _dcn_tiles(P, f"{b}.DyConv.{k}")
        wo = f32(b + ".offset.weight").permute(0, 2, 3, 1).reshape(27, -1)            # 27 -> 32 zero-padded rows
        P[b + ".offset.packed"] = torch.cat([wo, wo.new_zeros(5, wo.shape[1])], 0).to(dtype).contiguous()
        P[b + ".attn_w"] = f32(b + ".AttnConv.1.weight").reshape(-1)
        P[b + ".attn_b"] = f32(b + ".AttnConv.1.bias")
    for n in ("fpn_layer2", "fpn_layer3", "fpn_layer4", "top_blocks.p6", "top_blocks.p7"):
        w = f32(f"backbone.fpn.{n}.weight")
        P[f"backbone.fpn.{n}.packed"] = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(dtype).contiguous()
        _dcn_tiles(P, f"backbone.fpn.{n}")
    for n in ("fpn_inner2", "fpn_inner3", "fpn_inner4"):
        P[f"backbone.fpn.{n}.lin"] = h(f"backbone.fpn.{n}.weight").reshape(sd[f"backbone.fpn.{n}.weight"].shape[0], -1).contiguous()
    # patch embedding (4x4 stride-4 conv) as a GEMM over (kh, kw, c)-ordered patches; box / centerness 1x1 convs of every
    # level as ONE [8, 256] GEMM weight (4 box rows with the level's Scale folded, 1 centerness row, 3 zero rows): no
    # MIOpen call is left on the path (its solver choice -- down to a naive kernel -- varies from box to box)
    for l in range(5):
        s = f32(f"rpn.head.scales.{l}.scale")
        wb = (f32("rpn.head.bbox_pred.weight") * s).reshape(4, -1)
        wc = f32("rpn.head.centerness.weight").reshape(1, -1)
        P[f"rpn.head.boxctr.{l}.weight"] = torch.cat([wb, wc, wb.new_zeros(3, wb.shape[1])], 0).to(dtype).contiguous()
        P[f"rpn.head.boxctr.{l}.bias"] = torch.cat([f32("rpn.head.bbox_pred.bias") * s, f32("rpn.head.centerness.bias"),
                                                    wb.new_zeros(3)], 0).to(dtype).contiguous()
    # the same rows level-independent for mq_align_fused_fwd (the level's Scale is applied in its epilogue): [16, 256] + biases + scales
    wbc = torch.cat([f32("rpn.head.bbox_pred.weight").reshape(4, -1), f32("rpn.head.centerness.weight").reshape(1, -1)], 0)
    P["rpn.head.wbc"] = torch.cat([wbc, wbc.new_zeros(11, wbc.shape[1])], 0).to(dtype).contiguous()
    P["rpn.head.bbc"] = torch.cat([f32("rpn.head.bbox_pred.bias"), f32("rpn.head.centerness.bias"), wbc.new_zeros(3)], 0).contiguous()
    P["rpn.head.scales"] = torch.cat([f32(f"rpn.head.scales.{l}.scale").reshape(1) for l in range(5)], 0).contiguous()
    P["rpn.head.tok.weight"] = f32("rpn.head.dot_product_projection_text.weight")
    P["rpn.head.tok.bias"] = f32("rpn.head.dot_product_projection_text.bias")
    P["rpn.head.bias_lang32"] = f32("rpn.head.bias_lang")
    P["rpn.head.bias0_32"] = f32("rpn.head.bias0")
    P["rpn.head.inv_scale"] = float(math.exp(-float(sd["rpn.head.log_scale"])))
    for l in range(5):
        P[f"anchors.cell.{l}"] = f32(f"rpn.anchor_generator.cell_anchors.{l}")
    return P


def _ln(P, name, x, eps=1e-5, **kw):
    """LayerNorm through the HIP kernel (mq_layernorm_fwd); x fp16 or fp32 -> fp16 (see ops.layer_norm for **kw)."""
    return ops.layer_norm(x.contiguous(), P[name + ".weight"], P[name + ".bias"], eps, **kw)


def _add_ln(P, name, x, res, eps=1e-5, **kw):
    """LayerNorm(x + res) with the residual add fused into the kernel; res is the residual stream (fp32 or fp16)."""
    return ops.layer_norm(x.contiguous(), P[name + ".weight"], P[name + ".bias"], eps, residual=res.contiguous(), **kw)


def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


_NSPLIT_TARGET = int(os.environ.get("MQ_NSPLIT_TARGET", "768"))       # workgroups aimed at by the key split (tuning knob)


def _nsplit(n_blocks, n_key_tiles, target=None):
    """Split the key range when a launch would leave most of the 256 CUs idle."""
    target = target or _NSPLIT_TARGET
    if n_blocks >= 256 or n_key_tiles < 8:
        return 1
    return max(1, min(n_key_tiles // 4, -(-target // n_blocks), 32))


def _nsplit_t2i(n_img, heads, t_live, n_key_tiles, cus_per_xcd=32):
    """Key split of the VLFuse text side.  A group (image, split) is ceil(heads * ceil(t_live / 16) / 8) workgroups, each streaming
    ceil(tiles / nsplit) key tiles; one workgroup per CU; group g runs on XCD g % 8 (its workgroups stream the same tiles out of that
    XCD's L2).  Cost model: passes of an XCD's 32 CUs over the workgroups of ITS groups x (tiles per workgroup + ~6 tile-times of fixed
    work: Q load, partial write, its share of the merge) -- the split with the fewest tile-times wins.  141-token caption, 9 workgroups
    per group: B = 8 -> nsplit 7 (7 groups = 63 workgroups per XCD: 2 passes of 50 tiles); B = 4 -> nsplit 14 (the same 63 per XCD,
    25 tiles each: 0.151 ms against 0.231 ms for nsplit 7, whose 4 groups = 36 workgroups per XCD are 2 badly filled passes -- GPU call 10
    of round 4, profiles/r04_call10_t2i_sweep.json; round 3 counted passes of the whole chip and picked 7)."""
    if n_key_tiles < 8:
        return 1
    members = -(-(heads * (-(-t_live // 16))) // 8)
    best, best_cost = 1, None
    for ns in range(1, min(32, n_key_tiles // 4) + 1):
        per_xcd = -(-(n_img * ns) // 8) * members
        cost = (-(-per_xcd // cus_per_xcd)) * (-(-n_key_tiles // ns) + 6)
        if best_cost is None or cost < best_cost:
            best, best_cost = ns, cost
    return best


# ----------------------------------------------------------------------------- Swin + FPN
def swin_forward(P, cfg, img, p="backbone.body", SW=None):
    """swint.py:591-615 (and GroundingDINO's backbone/swin_transformer.py:688-741 with out_indices (1, 2, 3): same blocks,
    same names under prefix `p`).  img [B,3,H,W] fp16 -> [c3, c4, c5] as NHWC tensors (c2 is never used by the FPN,
    fpn.py:82-84, so its output norm / layout change is skipped)."""
    M = SW if SW is not None else cfg.MODEL.SWINT
    ws = M.WINDOW_SIZE
    _, _, H0, W0 = img.shape
    if W0 % 4 or H0 % 4:
        img = F.pad(img, (0, (4 - W0 % 4) % 4, 0, (4 - H0 % 4) % 4))
    B, Cin, Hi, Wi = img.shape
    H, W = Hi // 4, Wi // 4
    r32 = P["_r32"]
    h1_first = None
    nhwc = img.permute(0, 2, 3, 1)
    pe = ops.KERNELS["PATCH_EMBED_FUSED"] == 1 and r32 and Cin == 3 and (p + ".patch_embed.wpk") in P
    if pe and img.dtype == torch.float32 and img.is_contiguous():
        # projection + patch_embed.norm + norm1 of the first block in one pass over the caller's fp32 NCHW pixels (mq_patch_embed_fwd)
        x, h1_first = ops.patch_embed(img, P[p + ".patch_embed.wpk_nchw"], *P[p + ".patch_embed.f32"], eps=1e-5)
    elif pe and img.dtype != torch.float32 and nhwc.is_contiguous():
        x, h1_first = ops.patch_embed(nhwc, P[p + ".patch_embed.wpk"], *P[p + ".patch_embed.f32"], eps=1e-5)
    else:
        if img.dtype == torch.float32:
            nhwc = nhwc.to(P[p + ".patch_embed.lin"].dtype)
        patches = nhwc.reshape(B, H, 4, W, 4, Cin).permute(0, 1, 3, 2, 4, 5).reshape(B, H * W, 16 * Cin)
        x = F.linear(patches, P[p + ".patch_embed.lin"], P[p + ".patch_embed.proj.bias"])  # PatchEmbed.proj, swint.py:447-471
        # x is the residual stream of the stage: fp32 with RESIDUAL_FP32 (the LayerNorm kernel reads / writes it in fp32 and
        # hands fp16 to the GEMMs), fp16 otherwise
        x = _ln(P, p + ".patch_embed.norm", x, want_y=not r32, want_y32=r32)
    outs = []
    for i, (depth, heads) in enumerate(zip(M.DEPTHS, M.NUM_HEADS)):
        C = x.shape[-1]
        fused = P["_swin_fused_mlp"] and C in P["_swin_fused_widths"] and x.dtype == torch.float32
        pend = None                                    # MLP output whose residual add is fused into the next LayerNorm
        h1 = None                                      # norm1 of the current block when the previous fused MLP produced it
        if i == 0 and h1_first is not None:
            h1 = h1_first                              # ... or the patch-embedding kernel, for the very first block
        for j in range(depth):
            b = f"{p}.layers.{i}.blocks.{j}"
            shift = 0 if j % 2 == 0 else ws // 2
            if h1 is None:
                if pend is None:
                    h1 = _ln(P, b + ".norm1", x)
                else:
                    h1, x = _add_ln(P, b + ".norm1", pend, x)                               # x = x + mlp(...)  (swint.py:240)
            if ops.window_qkv_fused(C, ws, h1.numel()):
                a = ops.window_attention_qkv(h1.reshape(B, H, W, C), P[b + ".attn.qkv.weight"], P[b + ".attn.qkv.bias"],
                                             P[b + ".attn.rel_bias"], heads, ws, shift)               # the qkv tensor is never written
            else:
                qkv = _lin(P, b + ".attn.qkv", h1).reshape(B, H, W, 3 * C)
                a = ops.window_attention(qkv, P[b + ".attn.qkv.bias"], P[b + ".attn.rel_bias"], heads, ws, shift)
            proj = _lin(P, b + ".attn.proj", a.reshape(B, H * W, C))
            if fused:
                # one kernel: x += proj (swint.py:236); x += fc2(gelu(fc1(norm2(x)))) (:240); and the LayerNorm that reads
                # the result next (the following block's norm1, or the stage's output norm :611)
                nxt = f"{p}.layers.{i}.blocks.{j + 1}.norm1" if j + 1 < depth else (f"{p}.norm{i}" if i > 0 else None)
                nln = None if nxt is None else (P[nxt + ".weight"], P[nxt + ".bias"], 1e-5)
                r = ops.swin_mlp2(x.contiguous(), proj.contiguous(), P[b + ".norm2.weight"], P[b + ".norm2.bias"], 1e-5,
                                  P[b + ".mlp.w1f"], P[b + ".mlp.fc1.bias"], P[b + ".mlp.w2f"], P[b + ".mlp.fc2.bias"], next_ln=nln)
                x, h1 = r if nxt is not None else (r, None)
                pend = None
            else:
                h2, x = _add_ln(P, b + ".norm2", proj, x)                                   # x = x + attn(...)  (:236)
                pend = _lin(P, b + ".mlp.fc2", F.gelu(_lin(P, b + ".mlp.fc1", h2)))
                h1 = None
        if fused:
            if i > 0:
                outs.append(h1.reshape(B, H, W, C))                                         # = norm{i}(x), from the last fused MLP
        elif i > 0:
            o, x = _add_ln(P, f"{p}.norm{i}", pend, x)
            outs.append(o.reshape(B, H, W, C))
        else:
            x = x + pend
        if i < len(M.DEPTHS) - 1:                      # PatchMerging, swint.py:258-284
            d = f"{p}.layers.{i}.downsample"
            if ops.KERNELS["PATCH_MERGE_FUSED"] == 1:
                # gather + LayerNorm in one kernel (csrc/layernorm2.hip): no pad / cat pass; same values bit for bit
                yn = ops.patch_merge_ln(x.reshape(B, H, W, C).contiguous(), P[d + ".norm.weight"], P[d + ".norm.bias"], 1e-5)
                H, W = (H + 1) // 2, (W + 1) // 2
            else:
                y = x.reshape(B, H, W, C)
                if H % 2 or W % 2:
                    y = F.pad(y, (0, 0, 0, W % 2, 0, H % 2))
                y = torch.cat([y[:, 0::2, 0::2], y[:, 1::2, 0::2], y[:, 0::2, 1::2], y[:, 1::2, 1::2]], -1)
                H, W = (H + 1) // 2, (W + 1) // 2
                yn = _ln(P, d + ".norm", y.reshape(B, H * W, 4 * C))
            wr = P[d + ".reduction.weight"]
            if r32 and yn.is_cuda and yn.dtype != torch.float32 and os.environ.get("MQ_SWIN_RED_F32OUT") == "1":
                # A/B switch (GPU calls 26 / 27 of round 6): the reduction GEMM writing the fp32 residual stream itself instead of fp16 + a cast pass
                x = torch.mm(yn.reshape(-1, yn.shape[-1]), wr.t(), out_dtype=torch.float32).view(B, H * W, wr.shape[0])
            else:
                x = F.linear(yn, wr)
                if r32:
                    x = x.float()
    return outs


def fpn_forward(P, feats_nhwc):
    """fpn.py:59-129 + LastLevelP6P7 (:150-154).  NHWC in, NCHW-views of NHWC tensors out.  1x1 laterals are
    library GEMMs on the NHWC tokens, 3x3 convs the deterministic HIP implicit-GEMM kernel (mq_conv3x3_fwd)."""
    p = "backbone.fpn"
    c3, c4, c5 = feats_nhwc

    def lateral(name, x):
        return F.linear(x, P[f"{p}.{name}.lin"], P[f"{p}.{name}.bias"])

    def conv3(name, x, stride=1):
        return ops.conv3x3(x, P[f"{p}.{name}.packed"], P[f"{p}.{name}.bias"], P[f"{p}.{name}.packed"].shape[0], stride)
    via_dcn = ops.KERNELS["FPN_VIA_DCN"] == 1
    inner = lateral("fpn_inner4", c5)
    inners = [("fpn_layer4", inner)]
    for feat, idx in ((c4, 3), (c3, 2)):
        lat = lateral(f"fpn_inner{idx}", feat)
        if ops.KERNELS["FPN_TOPDOWN_FUSED"] == 1:
            inner = ops.add_upsample_nearest_(lat.contiguous(), inner.contiguous())           # one pass instead of three
        else:
            up = F.interpolate(inner.permute(0, 3, 1, 2), size=lat.shape[1:3], mode="nearest").permute(0, 2, 3, 1)
            inner = (lat + up).contiguous()
        inners.insert(0, (f"fpn_layer{idx}", inner))
    if via_dcn:
        # The three output convs (fpn.py:106-127) do not depend on each other: ONE grouped launch of the fused DCNv2 kernel with zero
        # offsets and mask logits of +100 (sigmoid = 1 exactly) -- a deformable conv sampling at integer positions with weights
        # (1, 0, 0, 0) IS the plain 3x3 conv (zero padding included: taps at -1 / H are "outside"), and that kernel runs its 128 x 256 x 64
        # tiles at 2.5x the rate of conv_igemm's 128 x 256 x 32 ones: +3.8 % end to end (round 3, GPU call 1).
        outs = ops.dcnv2_group([dict(x=x, om=_zero_offsets(x.shape[0], x.shape[1], x.shape[2], x.device), w=_dcn_w(P, f"{p}.{n}"),
                                     bias=P[f"{p}.{n}.bias"], stride=1, plain=True) for n, x in inners], want_stats=False, tag="dcnv2_fpn")
        res = [y.reshape(x.shape[0], hw[0], hw[1], 256) for (y, hw, _), (_, x) in zip(outs, inners)]

        def conv3s2(name, x):
            Ho, Wo = (x.shape[1] - 1) // 2 + 1, (x.shape[2] - 1) // 2 + 1
            y, hw = ops.dcnv2(x, _zero_offsets(x.shape[0], Ho, Wo, x.device), _dcn_w(P, f"{p}.{name}"), P[f"{p}.{name}.bias"], 2, tag="dcnv2_fpn", plain=True)
            return y.reshape(x.shape[0], hw[0], hw[1], 256)
        p6 = conv3s2("top_blocks.p6", res[-1])
        p7 = conv3s2("top_blocks.p7", F.relu(p6))
    else:
        res = [conv3(n, x) for n, x in inners]
        p6 = conv3("top_blocks.p6", res[-1], 2)
        p7 = conv3("top_blocks.p7", F.relu(p6), 2)
    return [t.permute(0, 3, 1, 2) for t in res + [p6, p7]]


_ZERO_OM = {}


def _zero_offsets(B, Ho, Wo, device):
    """[B, 27, Ho, Wo] fp32: 18 zero offsets + 9 mask logits of 100 (sigmoid(100) == 1.0f): the DCNv2 input that makes it a plain conv."""
    key = (B, Ho, Wo, str(device))
    if key not in _ZERO_OM:
        if len(_ZERO_OM) > 64:
            _ZERO_OM.clear()
        om = torch.zeros(B, 27, Ho, Wo, dtype=torch.float32, device=device)
        om[:, 18:] = 100.0
        _ZERO_OM[key] = om
    return _ZERO_OM[key]


def pooled_fpn_tokens(feats):
    """generalized_vl_rcnn_new.py:291-293 -> [B, sum(hw/4), C]."""
    if ops.KERNELS.get("POOLED_TOKENS_FUSED", 0) == 1 and ops.pool2x2_tokens_supported(feats):
        return ops.pool2x2_tokens(feats)                      # one launch: the same values bit for bit
    return torch.cat([F.avg_pool2d(f, 2).permute(0, 2, 3, 1).flatten(1, 2) for f in feats], 1)


# ----------------------------------------------------------------------------- language backbone
def bert_layer(P, b, x, key_bias, clamp, kv_len=None, x32=None, qk_mask=None, max_kv=0):
    """HF BertLayer / rpn/modeling_bert.py:71-272 (clamp=True): QK^T, mask, softmax, PV in one HIP kernel.
    x [B,T,C] fp16 (GEMM operand); x32: the same hidden state unrounded (fp32 residual stream) or None.
    Returns y16 (and y32 when x32 is given)."""
    Bn, T, C = x.shape
    r32 = x32 is not None
    if ops.KERNELS["BERT_ATTN_QKV_FUSED"] >= 1 and qk_mask is None and (b + ".qkv.frag") in P and ops.bert_attention_qkv_fits(T, C, 12, key_bias, batch=Bn):
        # projection + attention of every (batch item, head) in one launch: no qkv tensor (mq_bert_attn_qkv_fwd)
        ctx = ops.bert_attention_qkv(x, P[b + ".qkv.frag"], P[b + ".qkv.bias"], 12, key_bias=key_bias, clamp=50000.0 if clamp else 0.0, kv_len=kv_len, packed=True)
    elif ops.KERNELS["BERT_QKV_FUSED"] == 1 and qk_mask is None and ops.attention_text_fits(T, kv_len, max_kv) and (b + ".qkv.weight") in P \
            and (key_bias is None or key_bias.dim() == 2):
        ctx = ops.attention_text(_lin(P, b + ".qkv", x), 12, key_bias=key_bias, clamp=50000.0 if clamp else 0.0, kv_len=kv_len, max_kv=max_kv)
    else:
        qk = _lin(P, b + ".qk", x)                                                      # [B, T, 2C]
        vt = torch.baddbmm(P[b + ".attention.self.value.bias"][None, :, None], P[b + ".attention.self.value.weight"][None]
                           .expand(Bn, -1, -1), x.transpose(1, 2))                      # V^T [B, C, T]
        ctx = ops.attention(qk[:, :, :C], qk[:, :, C:], vt, 12, C // 12, key_bias=key_bias, clamp=50000.0 if clamp else 0.0,
                            kv_len=kv_len, qk_mask=qk_mask)
    a = _add_ln(P, b + ".attention.output.LayerNorm", _lin(P, b + ".attention.output.dense", ctx), x32 if r32 else x, 1e-12,
                want_sum=False, want_y32=r32)
    a16, a32 = a if r32 else (a, None)
    hmid = _lin(P, b + ".intermediate.dense", a16)
    if clamp and ops.KERNELS["BERT_CLAMP_FUSED"] == 1 and hmid.numel() % 8 == 0:
        # the five torch.clamp passes and the GELU of this half inside two kernels (equal results)
        o = _lin(P, b + ".output.dense", ops.clamp_gelu_clamp(hmid.contiguous(), 50000.0))
        return ops.layer_norm(o.contiguous(), P[b + ".output.LayerNorm.weight"], P[b + ".output.LayerNorm.bias"], 1e-12,
                              residual=(a32 if r32 else a16).contiguous(), want_sum=False, want_y32=r32, clamp=50000.0)
    if clamp:
        hmid = F.gelu(hmid.clamp(-50000, 50000)).clamp(-50000, 50000)
        o = _lin(P, b + ".output.dense", hmid).clamp(-50000, 50000)
    else:
        o = _lin(P, b + ".output.dense", F.gelu(hmid))
    y = _add_ln(P, b + ".output.LayerNorm", o, a32 if r32 else a16, 1e-12, want_sum=False, want_y32=r32)
    if not r32:
        return y.clamp(-50000, 50000) if clamp else y
    y16, y32 = y
    if clamp:
        y16, y32 = y16.clamp(-50000, 50000), y32.clamp(-50000, 50000)
    return y16, y32


def pre_select(P, p, vision, image, scale, side_ok=False):
    """modeling_bert_new.py:398-409,433-448: vision queries attend to the pooled image tokens (8 x 32).
    side_ok: the caller runs on the capturing / main stream -- the keys and values of the SECOND layer (they depend on the image tokens only)
    are projected on a side stream beside the first layer instead of inside the serial chain."""
    vision, image = vision * scale, image * scale
    Bn, Np, C = image.shape
    pad = (-Np) % 8
    if pad:
        image = F.pad(image, (0, 0, 0, pad))

    def image_kv(i):
        ic = f"{p}.layers.{i}.image_condition"
        kn = _ln(P, ic + ".norm_kv", image)
        return F.linear(kn, P[ic + ".to_k.weight"]), torch.matmul(P[ic + ".to_v.weight"], kn.transpose(1, 2))     # k, V^T [B, 256, Np_pad]
    ahead = None
    if side_ok and image.is_cuda:
        main, side = torch.cuda.current_stream(), _side_streams(image.device, 1, "pre_kv")[0]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ahead = image_kv(1)
    for i in range(2):
        b = f"{p}.layers.{i}"
        ic = b + ".image_condition"
        q = F.linear(_ln(P, ic + ".norm", vision), P[ic + ".to_q.weight"])
        if i == 1 and ahead is not None:
            main.wait_stream(side)
            k, vt = ahead
        else:
            k, vt = image_kv(i)
        nq_tiles = -(-q.shape[1] // 128)
        att = ops.attention(q, k, vt, 8, 32, nk=Np, nsplit=_nsplit(nq_tiles * Bn * 8, -(-Np // 64)))
        v16 = vision if vision.dtype == image.dtype else vision.to(image.dtype)
        res = F.linear(v16, P[b + ".res_mapping.weight"]) if (b + ".res_mapping.weight") in P else vision
        att_o = F.linear(att, P[ic + ".to_out.weight"])
        vision = att_o.float() + res.float() if P["_r32"] else att_o + res              # residual stream of the vision queries
        ff = b + ".ff"
        vision = vision + F.linear(F.gelu(F.linear(_ln(P, ff + ".norm", vision), P[ff + ".linear1.weight"])),
                                   P[ff + ".linear2.weight"])
    return vision


def gcp_kv(P, b, vision):
    """Keys | values of the vision queries for one GCP block: projected once per unique vision token; depend on `vision` only."""
    return F.linear(_ln(P, b + ".attn.norm_kv", vision), P[b + ".attn.to_kv.weight"])


def gcp_block(P, b, x, vision, idx, gates=None, kv=None):
    """GatedCrossAttentionBlock.forward (modeling_bert_new.py:298-374): K/V projected once per unique vision
    token, sparse gather-attention kernel, gate MLP + tanh + residual fused.  x: the text residual stream (fp32 with
    RESIDUAL_FP32, else fp16), returned in the same dtype; vision: fp16 or fp32; kv: gcp_kv(P, b, vision) computed ahead (side stream) or None."""
    if kv is None:
        kv = gcp_kv(P, b, vision)
    ff = b + ".ff"
    if ops.KERNELS["GCP_ATTN_FUSED"] >= 1 and (b + ".attn.to_q.frag") in P and ops.gcp_attention_fits(x, idx, policy=True):
        # LayerNorm, to_q, sparse attention, to_out, gate MLP, gated residual and the feed-forward half's LayerNorm: one launch (mq_gcp_attn_fwd)
        def ln(n):
            return (P[n + ".weight"], P[n + ".bias"])
        r = ops.gcp_attention(x.contiguous(), kv.contiguous(), idx, P[b + ".attn.to_q.frag"], P[b + ".attn.to_out.frag"], P[b + ".attn_gate.linear1.frag"],
                              P[b + ".attn_gate.w2"], ln(b + ".attn.norm"), ln(b + ".attn_gate.norm"), ln(ff + ".norm"), want_gate=gates is not None,
                              packed=True)
        x, xn = r[0], r[1]
        if gates is not None:
            gates.append(r[2])
        return x + F.linear(F.gelu(F.linear(xn, P[ff + ".linear1.weight"])), P[ff + ".linear2.gated"])
    q = F.linear(_ln(P, b + ".attn.norm", x), P[b + ".attn.to_q.weight"])
    sup = F.linear(ops.gcp_sparse_attention(q, kv, idx), P[b + ".attn.to_out.weight"])
    gh = F.linear(_ln(P, b + ".attn_gate.norm", sup), P[b + ".attn_gate.linear1.weight"])
    if gates is not None:
        x, g = ops.gcp_gate_residual(sup, gh, P[b + ".attn_gate.w2"], x.contiguous(), want_gate=True)
        gates.append(g)
    else:
        x = ops.gcp_gate_residual(sup, gh, P[b + ".attn_gate.w2"], x.contiguous())
    return x + F.linear(F.gelu(F.linear(_ln(P, ff + ".norm", x), P[ff + ".linear1.weight"])), P[ff + ".linear2.gated"])


def language_front(P, cfg, input_ids, attention_mask, use_vq, p="language_backbone.body.model", position_ids=None, qk_mask=None, max_kv=0):
    """Embeddings + the BERT layers that do not depend on the image (all 12 without vision queries, the first QV_START
    with them): the detector runs this on a side stream while the Swin backbone occupies the main one.
    MQ-GroundingDINO (prefix "bert"): `position_ids` [B,T] restart in every sub-sentence and `qk_mask` [B,1|H,T,T] uint8 is the
    block mask between special tokens (bertwarper.py:273-320); `attention_mask` is then None (no key-padding term: padding
    tokens see only themselves, exactly as in the reference)."""
    LB = cfg.MODEL.LANGUAGE_BACKBONE
    T = input_ids.shape[1]
    pe = P[p + ".embeddings.position_embeddings.weight"]
    e = P[p + ".embeddings.word_embeddings.weight"][input_ids].float() \
        + P[p + ".embeddings.token_type_embeddings.weight"][0].float() \
        + (pe[:T].float()[None] if position_ids is None else pe[position_ids].float())
    x32 = F.layer_norm(e, (e.shape[-1],), P[p + ".embeddings.LayerNorm.weight"].float(),
                       P[p + ".embeddings.LayerNorm.bias"].float(), 1e-12)
    x = x32.to(P[p + ".embeddings.LayerNorm.weight"].dtype)
    if not P["_r32"]:
        x32 = None
    key_bias = kv_len = None
    if attention_mask is not None:
        key_bias = ((1.0 - attention_mask.float()) * NEG).contiguous()
        # index of the last valid text token + 1: the attention kernels skip key tiles that hold padding only
        kv_len = (attention_mask.to(torch.int32) * torch.arange(1, T + 1, device=attention_mask.device, dtype=torch.int32)) \
            .amax(1).to(torch.int32).contiguous()
    nl, qv0 = LB.get("NUM_HIDDEN_LAYERS", 12), LB.get("QV_START", 6)
    n_front = qv0 if use_vq else nl
    hidden = []
    for i in range(n_front):
        x, x32 = _bert(P, f"{p}.encoder.layer.{i}", x, x32, key_bias, False, kv_len, qk_mask, max_kv=max_kv)
        hidden.append(x if x32 is None else x32)
    return {"x": x, "x32": x32, "hidden": hidden, "key_bias": key_bias, "kv_len": kv_len, "next": n_front, "qk_mask": qk_mask}


def _bert(P, b, x, x32, key_bias, clamp, kv_len, qk_mask=None, max_kv=0):
    """bert_layer on the (fp16 operand, fp32 stream or None) pair."""
    if x32 is None:
        return bert_layer(P, b, x, key_bias, clamp, kv_len, qk_mask=qk_mask, max_kv=max_kv), None
    return bert_layer(P, b, x, key_bias, clamp, kv_len, x32=x32, qk_mask=qk_mask, max_kv=max_kv)


def language_backbone(P, cfg, input_ids, attention_mask, vision, images, idx, want_gates=False, front=None, max_kv=0, side_ok=False):
    """bert_model_new.BertEncoder.forward (:39-104) over QVBertModel.forward (modeling_bert_new.py:690-848).
    `front`: result of language_front (computed concurrently with the image backbone); None -> computed here."""
    p = "language_backbone.body.model"
    LB = cfg.MODEL.LANGUAGE_BACKBONE
    use_vq = vision is not None
    if front is None:
        front = language_front(P, cfg, input_ids, attention_mask, use_vq, max_kv=max_kv)
    x, x32, hidden, key_bias, kv_len = front["x"], front.get("x32"), list(front["hidden"]), front["key_bias"], front["kv_len"]
    nl, qv0 = LB.get("NUM_HIDDEN_LAYERS", 12), LB.get("QV_START", 6)
    kv_ahead, kv_join = {}, None
    if use_vq:
        vision = pre_select(P, p + ".pre_select", vision, images, cfg.VISION_QUERY.VISION_SCALE, side_ok=side_ok)
        first = max(front["next"], qv0)
        if side_ok and vision.is_cuda and nl - first > 1:
            # K / V of the vision queries for GCP blocks 2 .. depend on `vision` only: off the serial text chain, onto a side stream (the first
            # block's stay in the chain: it needs them at once).  side_ok = the caller is on the capturing stream (no fork from a forked stream).
            main, side = torch.cuda.current_stream(), _side_streams(vision.device, 1, "gcp_kv")[0]
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for i in range(first + 1, nl):
                    kv_ahead[i] = gcp_kv(P, f"{p}.encoder.qv_layer.{i - qv0}", vision)
            kv_join = (main, side)
    gates = [] if want_gates else None
    for i in range(front["next"], nl):
        if use_vq and i >= qv0:
            if kv_join is not None and i in kv_ahead:
                kv_join[0].wait_stream(kv_join[1])
                kv_join = None
            if x32 is None:
                x = gcp_block(P, f"{p}.encoder.qv_layer.{i - qv0}", x, vision, idx, gates, kv=kv_ahead.get(i))
            else:
                x32 = gcp_block(P, f"{p}.encoder.qv_layer.{i - qv0}", x32, vision, idx, gates, kv=kv_ahead.get(i))
                x = x32.to(x.dtype)
        x, x32 = _bert(P, f"{p}.encoder.layer.{i}", x, x32, key_bias, False, kv_len, max_kv=max_kv)
        hidden.append(x if x32 is None else x32)
    n = LB.N_LAYERS
    # (one layer -- the shipped configs: mean over one tensor / 1 is the tensor; three launches of the serial language chain less)
    feats = hidden[-1].float() if n == 1 else torch.stack(hidden[-n:], 1).float().mean(1) / n
    m = attention_mask.unsqueeze(-1).float()
    embedded = feats * m
    aggregate = embedded.sum(1) / attention_mask.sum(-1, keepdim=True).float()
    return {"aggregate": aggregate, "embedded": embedded, "masks": attention_mask, "hidden": x, "hidden32": x32,
            "key_bias": key_bias, "kv_len": kv_len, "vision_query_gates": gates, "augmented_vision": vision}


# ----------------------------------------------------------------------------- VLDyHead
def _to_tokens(feats):
    """list of [B,C,H,W] (NHWC memory) -> ([B, N, C] token buffer, [(H, W)])."""
    sizes = [tuple(f.shape[-2:]) for f in feats]
    return torch.cat([f.permute(0, 2, 3, 1).flatten(1, 2) for f in feats], 1), sizes


def _level_views(tok, sizes):
    """[B, N, C] token buffer -> per-level [B, C, H, W] views (NHWC memory, batch stride N*C; no copies)."""
    Bn, N, C = tok.shape
    out, s = [], 0
    for (hh, ww) in sizes:
        out.append(tok[:, s:s + hh * ww].reshape(Bn, hh, ww, C).permute(0, 3, 1, 2))
        s += hh * ww
    return out


def vl_text_prep(P, b, hidden, key_bias, hidden32=None):
    """Text-only operands of one VLFuse layer (fuse_helper.py:221-231 with the image-side projections folded in, DESIGN.md
    section 4): LN(l), folded keys kf [B,8,T,256], folded values vo [B,8,T,256], per-(head, key) logit bias [B,8,T].
    hidden32: the text stream unrounded (fp32) or None; LN(l) is then kept in fp32 for the residual (fuse_helper.py:425)."""
    Bn = hidden.shape[0]
    if hidden32 is not None:
        l_ln, l_res = _ln(P, b + ".layer_norm_l", hidden32, want_y32=True)
    else:
        l_ln = l_res = _ln(P, b + ".layer_norm_l", hidden)
    T = l_ln.shape[1]
    C = P[b + ".Wq8"].shape[2]
    t = _lin(P, b + ".tprep", l_ln)                                                       # [B, T, 8*C | 8*C | 8 | pad]
    # [B, 8, T, 256] folded keys / values as VIEWS of the projection output (head stride 256, token stride = its row length): the VLFuse kernels
    # take the strides (ABI 30) -- round 5 made two permute().contiguous() copies per layer, 12 of the 13 layout copies of a step
    kf = t[..., :8 * C].unflatten(-1, (8, C)).permute(0, 2, 1, 3)
    vo = t[..., 8 * C:16 * C].unflatten(-1, (8, C)).permute(0, 2, 1, 3)
    bias = (t[..., 16 * C:16 * C + 8].float().permute(0, 2, 1) + key_bias[:, None, :]).contiguous()   # [B, 8, T] fp32
    return {"l_ln": l_ln, "l_res": l_res, "kf": kf, "vo": vo, "bias": bias}


def vl_image_side(P, b, v_ln, prep, kv_len=None, max_kv=0):
    """Image side of VLFuse: queries = LN(v) shared by the 8 heads; head sum, out-proj bias and the residual (on the
    NORMED v, fuse_helper.py:424) fused into the kernel."""
    return ops.vlfuse_i2t(v_ln, prep["kf"], prep["vo"], prep["bias"], P[b + ".ov.bias"], kv_len=kv_len, max_kv=max_kv)


def vl_text_side(P, b, v_ln, prep, kv_len=None, max_kv=0):
    """Text side of VLFuse: queries = folded text keys, keys = values = LN(v); values_v_proj, out_l_proj and gamma_l are
    one folded [768, 2048] weight applied to the result; residual on the NORMED l (fuse_helper.py:425)."""
    Bn, N, _ = v_ln.shape
    T = prep["l_ln"].shape[1]
    t_live = min(T, max_kv) if (kv_len is not None and max_kv > 0) else T             # 128-row tiles of pure padding are skipped
    out_l = ops.vlfuse_t2i(prep["kf"], v_ln, _nsplit_t2i(Bn, prep["kf"].shape[1], t_live, -(-N // 64)), kv_len=kv_len, max_kv=t_live)
    return prep["l_res"] + _lin(P, b + ".olc", out_l)                  # fp32 when the text stream is fp32


def vl_fuse_tokens(P, b, v, hidden, key_bias, kv_len=None, max_kv=0):
    """BiAttentionBlockForCheckpoint / BiMultiHeadAttention (fuse_helper.py:218-303,377-426) on the pyramid token
    buffer v [B, N, 256] (all levels concatenated, the layout the whole head keeps): one set of logits, softmax over text
    for the image side and over image tokens for the text side -- one launch each of the two VLFuse kernels
    (vlfuse_attn.hip); the logits are never materialised (reference: 3 x [B*8, 22400, 256] fp32 tensors) and, with the
    projections folded into the text-side operands, neither are the [B, 22400, 2048] q / value tensors: both kernels
    read LN(v) directly.  (Serial composition; vldyhead runs the text pieces on a side stream.)"""
    v_ln = ops.layer_norm(v, P[b + ".layer_norm_v.weight"], P[b + ".layer_norm_v.bias"], 1e-5)
    prep = vl_text_prep(P, b, hidden, key_bias)
    return vl_image_side(P, b, v_ln, prep, kv_len, max_kv), vl_text_side(P, b, v_ln, prep, kv_len, max_kv)


def vl_fuse(P, b, feats, hidden, key_bias, kv_len=None, max_kv=0):
    """List-of-levels form of vl_fuse_tokens."""
    v, sizes = _to_tokens(feats)
    v_new, l_new = vl_fuse_tokens(P, b, v.contiguous(), hidden, key_bias, kv_len, max_kv)
    return _level_views(v_new, sizes), l_new


_UP_W = {}


def _upsample_pool_weights(hs, ws, H, W, device):
    """Per-source-pixel weights (wy[hs], wx[ws]) such that  mean_{H x W}(bilinear_up(y)) == sum wy*wx*y
    (align_corners=True, F.upsample_bilinear semantics of vldyhead.py:224)."""
    key = (hs, ws, H, W, device)
    if key not in _UP_W:
        def axis(n_src, n_dst):
            w = torch.zeros(n_src, dtype=torch.float64)
            for o in range(n_dst):
                s = o * (n_src - 1) / (n_dst - 1) if n_dst > 1 else 0.0
                i0 = min(int(s), n_src - 1)
                i1 = min(i0 + 1, n_src - 1)
                l = s - i0
                w[i0] += 1 - l
                w[i1] += l
            return (w / n_dst).float().to(device)
        _UP_W[key] = (axis(hs, H), axis(ws, W))
    return _UP_W[key]


def dyconv_tokens(P, cfg, b, tok, sizes, defer_relu=False):
    """DyConv.forward (vldyhead.py:205-247) on the pyramid token buffer tok [B, N, 256] -> new buffer of the same shape.
      1. per level: 27-channel offset / mask conv (LDS-window kernel)                       -- five streams
      2. ALL DCNv2 branches of the layer (3 per level, 13 in total) in ONE grouped launch (dcn_fused.hip): gather + bilinear
         blend + MFMA + GroupNorm statistics; offsets of the CURRENT level are re-used for the level's three branches exactly
         like the reference (flat-index quirk handled inside the gather)
      3. per level: the fused HIP epilogue (GroupNorm affine, bilinear up-sampling of the level+1 branch, scale attention,
         branch mean, DYReLU) written straight into the level's slice of the output buffer     -- five streams
    defer_relu: return (buffer BEFORE DYReLU, coef [NL, B, 4, C]) -- the caller's next LayerNorm applies it (ops.dyrelu_layer_norm)."""
    G = cfg.MODEL.GROUP_NORM
    nl = len(sizes)
    Bn, N, C = tok.shape
    offs = [0]
    for (hh, ww) in sizes:
        offs.append(offs[-1] + hh * ww)
    lv = [tok[:, offs[l]:offs[l + 1]].reshape(Bn, sizes[l][0], sizes[l][1], C) for l in range(nl)]      # NHWC views
    out = torch.empty_like(tok)
    streams = bool(cfg.MODEL.DYHEAD.get("LEVEL_STREAMS", True) and tok.is_cuda and nl > 1)

    def fan_out(fn):
        """fn(level) for every level: the big level (75 % of the positions) on the main stream, the others on side streams
        (inside the HIP-graph capture: parallel graph branches); P5-P7 alone cannot fill 256 CUs."""
        if not streams:
            for l in range(nl):
                fn(l)
            return
        main = torch.cuda.current_stream()
        side = _side_streams(tok.device, max(1, min(nl - 1, _LEVEL_SIDE_STREAMS)))
        for s_ in side:
            s_.wait_stream(main)
        for l in range(1, nl):
            with torch.cuda.stream(side[(l - 1) % len(side)]):
                fn(l)
        fn(0)
        for s_ in side:
            main.wait_stream(s_)

    om = [None] * nl

    def offsets(lvl):
        om[lvl] = ops.conv3x3_nchw32(lv[lvl], P[b + ".offset.packed"], P[b + ".offset.bias"], 27)      # [B, 27, H, W] fp32
    if ops.KERNELS["OFFSET_CONV_VARIANT"] == 3 and ops.conv3x3_nchw32_group_supported(lv, 27):
        # one conv for every level (vldyhead.py:205-215): one launch over the tiles of all levels, no fork / join
        om = ops.conv3x3_nchw32_group(lv, P[b + ".offset.packed"], P[b + ".offset.bias"], 27)
    else:
        fan_out(offsets)

    branches, owner = [], []
    for lvl in range(nl):
        H, W = sizes[lvl]
        spec = [(1, lv[lvl], 1)]
        if lvl > 0:
            spec.append((2, lv[lvl - 1], 2))
        if lvl < nl - 1:
            spec.append((0, lv[lvl + 1], 1))
        for k, x_nhwc, stride in spec:
            Ho, Wo = (x_nhwc.shape[1] - 1) // stride + 1, (x_nhwc.shape[2] - 1) // stride + 1
            wy = wx = None
            if (Ho, Wo) != (H, W):
                wy, wx = _upsample_pool_weights(Ho, Wo, H, W, tok.device)
            branches.append({"x": x_nhwc, "om": om[lvl], "w": _dcn_w(P, f"{b}.DyConv.{k}"), "bias": P[f"{b}.DyConv.{k}.conv.bias"],
                             "stride": stride, "wy": wy, "wx": wx})
            owner.append((lvl, k, len(spec)))
    ys = ops.dcnv2_group(branches, want_stats=True)
    # GroupNorm affine x scale attention of every branch, one launch (the statistics came out of the DCN epilogue)
    coefs = ops.dyconv_coef_group(
        [{"sums": sums, "n": Ho * Wo, "gamma": P[f"{b}.DyConv.{k}.bn.weight"], "beta": P[f"{b}.DyConv.{k}.bn.bias"], "nbranches": nb}
         for (lvl, k, nb), (y, (Ho, Wo), sums) in zip(owner, ys)], P[b + ".attn_w"], P[b + ".attn_b"], G.NUM_GROUPS, G.EPSILON)

    relu_coef = torch.empty(nl, Bn, 4, C, dtype=torch.float32, device=tok.device) if defer_relu else None

    def epilogue(lvl):
        H, W = sizes[lvl]
        fused = [(y, coef, Ho, Wo) for (l2, k, nb), (y, (Ho, Wo), sums), coef in zip(owner, ys, coefs) if l2 == lvl]
        o = out[:, offs[lvl]:offs[lvl + 1]]
        _, pool = ops.dyconv_fuse(fused, H, W, out=o)
        rw = (P[b + ".relu.fc.0.weight"], P[b + ".relu.fc.0.bias"], P[b + ".relu.fc.2.weight"], P[b + ".relu.fc.2.bias"])
        if defer_relu:
            ops.dyrelu_coef(pool, H * W, *rw, out=relu_coef[lvl])
        else:
            ops.dyrelu_(o, pool, *rw)
    if ops.KERNELS["DYCONV_EPILOGUE_GROUPED"] == 1 and C == 256 and nl <= 8:
        # all levels in two launches on this stream: fuse (one work list), DYReLU coefficients (grid B x levels)
        rw = (P[b + ".relu.fc.0.weight"], P[b + ".relu.fc.0.bias"], P[b + ".relu.fc.2.weight"], P[b + ".relu.fc.2.bias"])
        rc = relu_coef if defer_relu else torch.empty(nl, Bn, 4, C, dtype=torch.float32, device=tok.device)
        ops.dyconv_epilogue_group(
            [([(y, coef, Ho, Wo) for (l2, k, nb), (y, (Ho, Wo), sums), coef in zip(owner, ys, coefs) if l2 == lvl], sizes[lvl][0], sizes[lvl][1],
              out[:, offs[lvl]:offs[lvl + 1]]) for lvl in range(nl)], *rw, rc)
        if not defer_relu:
            fan_out(lambda lvl: ops.dyrelu_apply_(out[:, offs[lvl]:offs[lvl + 1]], rc[lvl]))
    else:
        fan_out(epilogue)
    return (out, relu_coef) if defer_relu else out


def dyconv(P, cfg, b, feats):
    """List-of-levels form of dyconv_tokens."""
    tok, sizes = _to_tokens(feats)
    return _level_views(dyconv_tokens(P, cfg, b, tok.contiguous(), sizes), sizes)


_SIDE_STREAMS = {}
_LEVEL_SIDE_STREAMS = int(os.environ.get("MQ_LEVEL_SIDE_STREAMS", "4"))      # side streams of the per-level DyConv work (P4 .. P7; P3 runs on the main one)


# priority of the side streams by tag (A/B switch MQ_STREAM_PRIORITY = "text:-1,levels:0": -1 = high): the text chain of a fusion layer is
# ~25 small launches that the next layer's image-side attention waits for; its kernels queue behind the workgroups of the DCNv2 launch
_STREAM_PRIORITY = dict((kv.split(":")[0], int(kv.split(":")[1])) for kv in os.environ.get("MQ_STREAM_PRIORITY", "").split(",") if ":" in kv)


def _side_streams(device, n, tag="levels"):
    key = (device.index, n, tag)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=device, priority=_STREAM_PRIORITY.get(tag, 0)) for _ in range(n)]
    return _SIDE_STREAMS[key]


def vldyhead(P, cfg, feats, lang, trace=None):
    """VLDyHead.forward (vldyhead.py:769-900), eval outputs.  The pyramid lives in ONE token buffer [B, N, 256] (levels
    concatenated) from the first fusion layer to the prediction heads: VLFuse reads / writes it whole, DyConv reads
    per-level NHWC views of it and writes per-level slices -- no concatenation or split copies between layers.

    Two-stream schedule per fusion layer (text work never waits for image work it does not need, and vice versa):
        main : LN(v) -> image-side attention -> DyConv (itself forked over the five levels)
        text : text-side attention (needs LN(v)) -> folded out-projection -> BERT layer -> operands of the NEXT layer
    The text chain (~0.8 ms of small or tail-heavy launches) hides under the image chain (~1.3 ms).
    trace: optional list; receives per fusion layer {"fuse_tok", "fuse_hidden", "bert_hidden", "dyconv_tok"} (parity ladder,
    single-stream schedule)."""
    p = "rpn.head"
    hidden, key_bias, kv_len = lang["hidden"], lang["key_bias"], lang.get("kv_len")
    h32 = lang.get("hidden32")
    max_kv = lang.get("max_kv", 0)
    tok, sizes = _to_tokens(feats)
    tok = tok.contiguous()
    L = cfg.MODEL.DYHEAD.NUM_CONVS
    t = f"{p}.dyhead_tower"
    two_streams = bool(tok.is_cuda and cfg.MODEL.DYHEAD.get("LEVEL_STREAMS", True)) and trace is None

    def text_side(b, v_ln, prep):
        """-> (hidden16, hidden32 or None) after the text half of VLFuse"""
        hnew = vl_text_side(P, b, v_ln, prep, kv_len, max_kv)
        return (hnew.to(tok.dtype), hnew) if h32 is not None else (hnew, None)

    # DYReLU of layers 0 .. L-2: applied by the next layer's layer_norm_v, the only reader of those buffers (the parity ladder wants
    # every layer's output: a trace run applies it to a COPY for the record and still takes the deferred path itself)
    defer = ops.KERNELS["DYRELU_IN_LN"] == 1 and tok.shape[-1] == 256 and len(sizes) <= 8
    relu_coef = None

    def relu_applied_copy():
        if relu_coef is None:
            return tok
        cp, off = tok.clone(), 0
        for l, (hh, ww) in enumerate(sizes):
            ops.dyrelu_apply_(cp[:, off:off + hh * ww], relu_coef[l])
            off += hh * ww
        return cp

    def norm_v(b):
        if relu_coef is not None:
            return ops.dyrelu_layer_norm(tok, relu_coef, sizes, P[b + ".layer_norm_v.weight"], P[b + ".layer_norm_v.bias"], 1e-5)
        return ops.layer_norm(tok, P[b + ".layer_norm_v.weight"], P[b + ".layer_norm_v.bias"], 1e-5)

    def dyconv_layer(i):
        if defer and i + 1 < L:
            return dyconv_tokens(P, cfg, f"{t}.{3 * i + 2}", tok, sizes, defer_relu=True)
        return dyconv_tokens(P, cfg, f"{t}.{3 * i + 2}", tok, sizes), None

    if not two_streams:
        for i in range(L):
            b = f"{t}.{3 * i}.b_attn"
            v_ln = norm_v(b)
            prep = vl_text_prep(P, b, hidden, key_bias, h32)
            tok = vl_image_side(P, b, v_ln, prep, kv_len, max_kv)
            hidden, h32 = text_side(b, v_ln, prep)
            rec = {"fuse_tok": tok, "fuse_hidden": hidden if h32 is None else h32}
            hidden, h32 = _bert(P, f"{t}.{3 * i + 1}", hidden, h32, key_bias, True, kv_len, max_kv=max_kv)
            tok, relu_coef = dyconv_layer(i)
            if trace is not None:
                rec.update(bert_hidden=hidden if h32 is None else h32, dyconv_tok=relu_applied_copy())
                trace.append(rec)
    else:
        main, text = torch.cuda.current_stream(), _side_streams(tok.device, 1, "text")[0]
        text.wait_stream(main)
        with torch.cuda.stream(text):
            prep = vl_text_prep(P, f"{t}.0.b_attn", hidden, key_bias, h32)
        keep = []                                             # main-stream tensors the text stream still reads
        for i in range(L):
            b = f"{t}.{3 * i}.b_attn"
            v_ln = norm_v(b)
            main.wait_stream(text)                            # operands of this layer ready; previous text chain done
            keep.clear()                                      # ... so last layer's LN(v) may be recycled now
            keep.append(v_ln)
            text.wait_stream(main)                            # LN(v) ready
            cur = prep
            with torch.cuda.stream(text):
                hidden, h32 = text_side(b, v_ln, cur)
                hidden, h32 = _bert(P, f"{t}.{3 * i + 1}", hidden, h32, key_bias, True, kv_len, max_kv=max_kv)
                if i + 1 < L:
                    prep = vl_text_prep(P, f"{t}.{3 * (i + 1)}.b_attn", hidden, key_bias, h32)
            tok = vl_image_side(P, b, v_ln, cur, kv_len, max_kv)
            tok, relu_coef = dyconv_layer(i)
        main.wait_stream(text)
        keep.clear()
    emb = F.normalize((hidden if h32 is None else h32).float(), p=2, dim=-1)
    tk = F.linear(emb / 2.0, P[p + ".tok.weight"], P[p + ".tok.bias"]) * P[p + ".inv_scale"]       # [B, T, 256]
    tbias = (emb @ P[p + ".bias_lang32"] + P[p + ".bias0_32"]).contiguous()                           # [B, T]
    live = min(tk.shape[1], max_kv) if max_kv > 0 else tk.shape[1]
    # (precise mode on the device: the text tile of the fused kernel is 264 floats per live token -- captions of up to 144 tokens fit the LDS)
    fits = ops.f32_operands() != 1 or -(-live // 16) * 16 <= 144
    if ops.KERNELS["ALIGN_FUSED"] == 1 and tk.shape[1] <= 256 and tok.shape[-1] == 256 and len(sizes) <= 8 and fits:
        # (shapes the fused kernel does not take -- more than 256 text tokens, other widths, more than 8 levels -- use the GEMM path below)
        # heads + alignment + scoring happen in ONE kernel inside postprocess() (mq_align_fused_fwd): hand over its operands
        return {"tok": tok, "sizes": sizes, "tk16": tk.to(tok.dtype).contiguous(), "tbias": tbias, "max_kv": max_kv,
                "wbc": P[p + ".wbc"], "bbc": P[p + ".bbc"], "scales": P[p + ".scales"],
                "feats": _level_views(tok, sizes), "hidden": hidden if h32 is None else h32}
    tok16_t = tk.to(tok.dtype).transpose(1, 2)
    Bn, N, C = tok.shape
    dots_all = torch.bmm(tok, tok16_t)                                                               # [B, N, T], all levels
    bbox, ctr, dots = [], [], []
    off = 0
    for l, (H, W) in enumerate(sizes):
        tokens = tok[:, off:off + H * W]
        bc = F.linear(tokens, P[f"{p}.boxctr.{l}.weight"], P[f"{p}.boxctr.{l}.bias"]).reshape(Bn, H, W, 8)
        bbox.append(bc[..., :4].permute(0, 3, 1, 2))                                                 # [B, 4, H, W] view
        ctr.append(bc[..., 4:5].permute(0, 3, 1, 2))
        dots.append(dots_all[:, off:off + H * W])                                                    # [B, HW, T] view
        off += H * W
    return {"bbox_reg": bbox, "centerness": ctr, "dot": dots, "tbias": tbias, "feats": _level_views(tok, sizes),
            "hidden": hidden if h32 is None else h32}


# ----------------------------------------------------------------------------- anchors + post-processing
def grid_anchors(P, sizes, strides, device):
    """anchor_generator.py:73-95."""
    out = []
    for l, ((H, W), s) in enumerate(zip(sizes, strides)):
        sx = torch.arange(0, W * s, step=s, dtype=torch.float32, device=device)
        sy = torch.arange(0, H * s, step=s, dtype=torch.float32, device=device)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        sh = torch.stack([xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)], 1)
        out.append((sh[:, None, :] + P[f"anchors.cell.{l}"][None]).reshape(-1, 4).contiguous())
    return out


TIE_SLOTS = 16      # output slots behind DETECTIONS_PER_IMG for detections tied with the last kept score; cfg.MODEL.ATSS.TIE_SLOTS overrides.
                    # More ties than slots are cut AND flagged: `tie_overflow` [B] in the result (bit 16 of the packed counts the detector reads)


def postprocess(cfg, head, anchors, im_wh, tokidx, label_ids, want_cls=False, level_streams=None):
    """ATSSPostProcessor.forward (rpn/inference.py:620-769) without per-image Python loops or host syncs:
    fixed-shape top-k per level, one sort, device-side NMS, fixed-shape top-(`DETECTIONS_PER_IMG` + TIE_SLOTS).
    Returns boxes [B,K2,4], scores [B,K2] (<= 0 => empty slot), labels [B,K2], counts [B] -- all on device; live slots are contiguous
    from slot 0 (scores sorted descending, ties behind slot K - 1 directly follow it)."""
    A = cfg.MODEL.ATSS
    dev = head["tbias"].device
    Bn = head["tbias"].shape[0]
    L = tokidx.shape[-2]
    if not torch.is_tensor(im_wh):                       # list of (h, w) -> [B, 2] (w, h)
        im_wh = torch.tensor([[w, h] for (h, w) in im_wh], dtype=torch.float32, device=dev)
    agg = ops.SCORE_AGG[str(cfg.MODEL.DYHEAD.get("SCORE_AGG", "MEAN")).upper()]
    fused = None
    if "tok" in head:
        # prediction heads, region-word alignment and per-location scoring of ALL levels: one launch, logits never written
        fused = ops.align_fused(head["tok"], head["tk16"], head["tbias"], head["wbc"], head["bbc"], head["scales"], tokidx, head["sizes"],
                                A.INFERENCE_TH, agg=agg, kv_max=head.get("max_kv", 0), want_cls=want_cls, want_logits=want_cls)
        hws = [h * w for (h, w) in head["sizes"]]
        if want_cls:                                     # raw mode: the reference's head outputs as views, for the parity ladder
            T = head["tbias"].shape[1]
            offs_ = [0]
            for n_ in hws:
                offs_.append(offs_[-1] + n_)
            head["dot"] = [fused["logits"][:, offs_[l]:offs_[l + 1]] for l in range(len(hws))]                        # without the bias
            head["bbox_reg"] = [fused["reg"][l].reshape(Bn, h, w, 4).permute(0, 3, 1, 2) for l, (h, w) in enumerate(head["sizes"])]
            head["centerness"] = [fused["ctr"][:, offs_[l]:offs_[l + 1]].reshape(Bn, 1, h, w) for l, (h, w) in enumerate(head["sizes"])]
    else:
        hws = [d.shape[1] for d in head["dot"]]
    ks = [min(A.PRE_NMS_TOP_N, hw * L) for hw in hws]
    tot = sum(ks)
    Kd = int(A.DETECTIONS_PER_IMG)
    K = min(Kd, tot) if Kd > 0 else tot
    K2 = min(K + int(A.get("TIE_SLOTS", TIE_SLOTS)), tot) if Kd > 0 else tot
    if ops.KERNELS["POST_FUSED"] == 1 and tot <= 16384 and (Kd <= 0 or tot <= 6656) and all(a.shape[0] == hw for a, hw in zip(anchors, hws)) \
            and ops.post_select_supported(hws, ks, Bn, L):
        # ---- csrc/post2.hip: exact per-level select + decode (slices, then levels), merge of the sorted level lists, NMS, final selection
        if fused is not None:
            ranked_l, reg_l = fused["ranked"], fused["reg"]
            cls_all = fused["cls"] if want_cls else None
        else:
            ranked_l, reg_l, cls_all = [], [], []
            for l, HW in enumerate(hws):
                ctr_flat = head["centerness"][l].permute(0, 2, 3, 1).reshape(Bn, HW).contiguous()
                r = ops.align_scores(head["dot"][l], head["tbias"], tokidx, ctr_flat, A.INFERENCE_TH, want_cls=want_cls, agg=agg)
                if want_cls:
                    r, c_ = r
                    cls_all.append(c_)
                ranked_l.append(r.float().contiguous())
                reg_l.append(head["bbox_reg"][l].permute(0, 2, 3, 1).reshape(Bn, HW, 4).float().contiguous())
        lab32 = label_ids if label_ids.dtype == torch.int32 else label_ids.to(torch.int32)
        wh32 = im_wh if (im_wh.dtype == torch.float32 and im_wh.is_contiguous()) else im_wh.float().contiguous()
        ub, us, ul, uid = ops.post_select(ranked_l, reg_l, [a.contiguous() for a in anchors], ks, lab32.contiguous(), wh32)
        boxes, scores, labels, nvalid = ops.post_sort(ub, us, ul, ks)
        keep8 = ops.ml_nms(boxes, labels, nvalid, A.NMS_TH, max_keep=K2 if Kd > 0 else 0, as_bool=False)
        packed, cnt = ops.post_finalize(boxes, scores, labels, keep8, K, K2)
        out = {"boxes": packed[..., :4], "scores": packed[..., 4], "labels": packed[..., 5].to(torch.int64), "counts": cnt & 0xFFFF,
               "tie_overflow": (cnt >> 16) > 0, "packed": packed, "counts_packed": cnt,
               "pre_nms": {"boxes": boxes, "scores": scores, "labels": labels, "nvalid": nvalid, "keep": keep8.bool()}}
        if want_cls:
            out["cls"] = cls_all
        return out
    boxes = torch.empty(Bn, tot, 4, dtype=torch.float32, device=dev)
    scores = torch.empty(Bn, tot, dtype=torch.float32, device=dev)
    labels = torch.empty(Bn, tot, dtype=torch.int32, device=dev)
    cls_all = [None] * len(ks)
    offs = [0]
    for k in ks:
        offs.append(offs[-1] + k)

    def level(l):
        anc, k, HW = anchors[l], ks[l], hws[l]
        if fused is not None:
            r, reg_nhwc = fused["ranked"][l], fused["reg"][l]
            if want_cls:
                cls_all[l] = fused["cls"][l]
        else:
            dot, reg, ctr = head["dot"][l], head["bbox_reg"][l], head["centerness"][l]
            ctr_flat = ctr.permute(0, 2, 3, 1).reshape(Bn, HW).contiguous()
            r = ops.align_scores(dot, head["tbias"], tokidx, ctr_flat, A.INFERENCE_TH, want_cls=want_cls, agg=agg)
            if want_cls:
                r, cls_all[l] = r
            reg_nhwc = reg.permute(0, 2, 3, 1).reshape(Bn, HW, 4).contiguous()
        val, flat = torch.topk(r.reshape(Bn, HW * L), k, dim=1, sorted=False)
        ops.box_decode(val.contiguous(), flat.contiguous(), reg_nhwc, anc, label_ids, im_wh, boxes, scores, labels, HW, L, offs[l])

    # the five levels are independent until the sort: one HIP stream each (the small levels are pure launch latency)
    if level_streams is None:
        level_streams = cfg.MODEL.DYHEAD.get("LEVEL_STREAMS", True)
    if boxes.is_cuda and level_streams and len(ks) > 1:
        main = torch.cuda.current_stream()
        side = _side_streams(dev, len(ks) - 1)
        for s_ in side:
            s_.wait_stream(main)
        for l in range(1, len(ks)):
            with torch.cuda.stream(side[l - 1]):
                level(l)
        level(0)
        for s_ in side:
            main.wait_stream(s_)
    else:
        for l in range(len(ks)):
            level(l)
    order = torch.argsort(scores, dim=1, descending=True, stable=True)
    scores = torch.gather(scores, 1, order)
    labels = torch.gather(labels, 1, order.to(torch.int64)).contiguous()
    boxes = torch.gather(boxes, 1, order[:, :, None].expand(-1, -1, 4)).contiguous()
    nvalid = (scores > 0).sum(1).to(torch.int32)
    # Final selection, rpn/inference.py:757-766: with more than DETECTIONS_PER_IMG survivors the reference keeps every detection whose
    # score is >= the K-th best one (torch.kthvalue + `>=`) -- detections TIED with the K-th are all kept, so an image can return more
    # than K.  Fixed shapes here: K + TIE_SLOTS output slots; the slots behind K are live only for scores equal to the K-th.
    keep = ops.ml_nms(boxes, labels, nvalid, A.NMS_TH, max_keep=K2 if Kd > 0 else 0)
    kept_scores = torch.where(keep, scores, torch.full_like(scores, -1.0))
    top, ti = torch.topk(kept_scores, K2, dim=1, sorted=True)
    if K2 > K:
        tie = top[:, K:] == top[:, K - 1:K]
        top = torch.cat([top[:, :K], torch.where(tie, top[:, K:], torch.full_like(top[:, K:], -1.0))], 1)
    # every tie slot taken by a score equal to the K-th: the NMS sweep stopped at K2 kept boxes, more ties may exist behind them (the
    # reference would return them all) -- detectable, not silent
    overflow = ((top[:, K2 - 1] == top[:, K - 1]) & (top[:, K - 1] > 0)) if 0 < K < K2 < tot else torch.zeros(Bn, dtype=torch.bool, device=dev)
    out = {"boxes": torch.gather(boxes, 1, ti[:, :, None].expand(-1, -1, 4)), "scores": top,
           "labels": torch.gather(labels, 1, ti).to(torch.int64), "counts": (top > 0).sum(1), "tie_overflow": overflow,
           "pre_nms": {"boxes": boxes, "scores": scores, "labels": labels, "nvalid": nvalid, "keep": keep}}
    if want_cls:
        out["cls"] = cls_all
    return out
