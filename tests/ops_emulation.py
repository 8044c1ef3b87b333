"""TEST-ONLY pure-torch emulations of the mq_det_amd.ops entry points (same signatures / layouts).

Purpose: validate, on the CPU-only build box, all the HOST-SIDE glue of the product pipeline (weight packing and
folding, NHWC layouts, strides, padding conventions, index construction, post-processing plumbing) against the
oracle BEFORE spending GPU minutes -- tests/test_pipeline_glue_cpu.py monkeypatches these into
mq_det_amd.modeling.pipeline.ops.  They are never imported by the product; the real kernels are tested
on the GPU by tests/test_gpu_parity.py.
"""
import math

import torch
import torch.nn.functional as F


def attention(q, k, vt, num_heads, head_dim, key_bias=None, scale=None, clamp=0.0, nsplit=1, nk=None, kv_len=None, qk_mask=None):
    B, Nq, HD = q.shape
    Nk = k.shape[1] if nk is None else nk
    H, D = num_heads, head_dim
    assert vt.shape[2] % 8 == 0 and vt.shape[2] >= Nk and q.stride(2) == 1 and k.stride(2) == 1
    qh = q.float().reshape(B, Nq, H, D).transpose(1, 2)
    kh = k.float()[:, :Nk].reshape(B, Nk, H, D).transpose(1, 2)
    vh = vt.float()[:, :, :Nk].reshape(B, H, D, Nk).transpose(2, 3)
    s = qh @ kh.transpose(-1, -2) * (scale if scale is not None else 1.0 / math.sqrt(D))
    if clamp > 0:
        s = s.clamp(-clamp, clamp)
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    if qk_mask is not None:
        assert qk_mask.shape == (B, H, Nq, Nk) and qk_mask.dtype in (torch.bool, torch.uint8)
        s = s.masked_fill(qk_mask.bool(), -1e30)
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, Nq, HD).to(q.dtype)


def patch_embed(img, wpk, bias, g0, b0, g1, b1, eps=1e-5):
    C = wpk.shape[0]
    if img.dtype == torch.float32:                         # [B, 3, Hi, Wi] fp32, rounded to the operand type; k = (channel, py, px)
        x = img.to(wpk.dtype).float()
        w4 = wpk.float()[:, :48].reshape(C, 3, 4, 4)
    else:                                                  # [B, Hi, Wi, 3] 16-bit; k = (py, px, channel)
        x = img.float().permute(0, 3, 1, 2)
        w4 = wpk.float()[:, :48].reshape(C, 4, 4, 3).permute(0, 3, 1, 2)
    y = F.conv2d(x, w4, bias, stride=4).flatten(2).transpose(1, 2)
    x32 = F.layer_norm(y, (C,), g0, b0, eps)
    return x32, F.layer_norm(x32, (C,), g1, b1, eps).to(wpk.dtype)


def attention_text(qkv, heads, key_bias=None, clamp=0.0, kv_len=None, max_kv=0, scale=None):
    B, T, C3 = qkv.shape
    HD = C3 // 3
    q, k, v = qkv[..., :HD], qkv[..., HD:2 * HD], qkv[..., 2 * HD:]
    return attention4(q.reshape(B, T, heads, -1), k.reshape(B, T, heads, -1), v.reshape(B, T, heads, -1).permute(0, 2, 3, 1).contiguous(),
                      key_bias=key_bias, scale=scale, clamp=clamp)


def bert_attention_qkv(x, wqkv, bqkv, heads, key_bias=None, clamp=0.0, kv_len=None, scale=None, packed=False):
    """ops.bert_attention_qkv: the fused projection (rounded to the operand type like the GEMM's output) + attention_text."""
    import torch.nn.functional as F
    from mq_det_amd.ops import unpack_b_fragments
    wqkv = unpack_b_fragments(wqkv) if packed else wqkv
    qkv = F.linear(x.float(), wqkv.float(), bqkv.float()).to(x.dtype)
    return attention_text(qkv, heads, key_bias=key_bias, clamp=clamp, kv_len=kv_len, scale=scale)


def attention4(q4, k4, vt4, key_bias=None, scale=None, clamp=0.0, nsplit=1, nk=None, kv_len=None):
    B, Nq, H, D = q4.shape
    Nk = k4.shape[1] if nk is None else nk
    assert vt4.shape[3] % 8 == 0 and q4.stride(3) == 1 and k4.stride(3) == 1 and vt4.stride(3) == 1
    for t in (q4, k4, vt4):
        assert all(st % 8 == 0 for st in t.stride()[:-1])
    qh = q4.float().permute(0, 2, 1, 3)
    kh = k4.float()[:, :Nk].permute(0, 2, 1, 3)
    vh = vt4.float()[..., :Nk].transpose(2, 3)
    s = qh @ kh.transpose(-1, -2) * (scale if scale is not None else 1.0 / math.sqrt(D))
    if key_bias is not None:
        kb = key_bias if key_bias.dim() == 3 else key_bias[:, None, :]
        masked = kb < -1e29
        s = s + torch.where(masked, torch.zeros_like(kb), kb)[:, :, None, :]
    if clamp > 0:
        s = s.clamp(-clamp, clamp)
    if key_bias is not None:
        s = s.masked_fill(masked[:, :, None, :].expand_as(s), -1e30)
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, Nq, H * D).to(q4.dtype)


def window_attention(qkv, qkv_bias, rel_bias, heads, ws, shift):
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    Hp, Wp = H + pad_b, W + pad_r
    full = qkv_bias.float().expand(B, Hp, Wp, C3).clone()
    full[:, :H, :W] = qkv.float()
    if shift:
        full = torch.roll(full, (-shift, -shift), (1, 2))
    xw = full.reshape(B, Hp // ws, ws, Wp // ws, ws, C3).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C3)
    Bw, N, _ = xw.shape
    t = xw.reshape(Bw, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    attn = (t[0] * (C // heads) ** -0.5) @ t[1].transpose(-1, -2) + rel_bias[None, :, :N, :N]
    if shift:
        region = torch.zeros(Hp, Wp)
        k = 0
        for h0, h1 in ((0, Hp - ws), (Hp - ws, Hp - shift), (Hp - shift, Hp)):
            for w0, w1 in ((0, Wp - ws), (Wp - ws, Wp - shift), (Wp - shift, Wp)):
                region[h0:h1, w0:w1] = k
                k += 1
        r = region.reshape(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, N)
        m = (r[:, None, :] != r[:, :, None]).float() * -100.0
        nW = m.shape[0]
        attn = (attn.reshape(B, nW, heads, N, N) + m[None, :, None]).reshape(Bw, heads, N, N)
    o = (attn.softmax(-1) @ t[2]).transpose(1, 2).reshape(Bw, N, C)
    o = o.reshape(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    return o[:, :H, :W].contiguous().to(qkv.dtype)


def window_attention_qkv(x, w, bias, rel_bias, heads, ws, shift):
    """Plain-torch statement of mq_window_attn_qkv_fwd: the qkv Linear (output rounded to the activation dtype, as the reference's
    tensor is) followed by the window attention restatement."""
    qkv = F.linear(x.float(), w.float(), bias.float()).to(x.dtype)
    return window_attention(qkv, bias, rel_bias, heads, ws, shift)


def gcp_sparse_attention(q, kv, idx, heads=8, dim_head=64):
    B, T, HD = q.shape
    S = idx.shape[2]
    out = torch.zeros_like(q, dtype=torch.float32)
    k, v = kv.float()[..., :HD], kv.float()[..., HD:]
    for b in range(B):
        for t in range(T):
            ids = [int(i) for i in idx[b, t] if int(i) >= 0]
            if not ids:
                continue
            qq = q[b, t].float().reshape(heads, dim_head) * dim_head ** -0.5
            kk = k[b, ids].reshape(len(ids), heads, dim_head)
            vv = v[b, ids].reshape(len(ids), heads, dim_head)
            w = torch.einsum("hd,shd->hs", qq, kk).softmax(-1)
            out[b, t] = torch.einsum("hs,shd->hd", w, vv).reshape(-1)
    return out.to(q.dtype)


def gcp_attention(x, kv, idx, wq, wout, wg1, w2, ln_a, ln_g, ln_f=None, eps=1e-5, want_gate=False, rows_per_block=0, packed=False):
    """ops.gcp_attention: the unfused chain with its rounding points (every LayerNorm / GEMM output rounded to the operand type once)."""
    dt = kv.dtype
    if packed:
        from mq_det_amd.ops import unpack_b_fragments
        wq, wout, wg1 = (unpack_b_fragments(w) for w in (wq, wout, wg1))

    def ln(v, gb):
        return F.layer_norm(v.float(), (v.shape[-1],), gb[0].float(), gb[1].float(), eps).to(dt)
    q = F.linear(ln(x, ln_a).float(), wq.float()).to(dt)
    sup = F.linear(gcp_sparse_attention(q, kv, idx).float(), wout.float()).to(dt)
    gh = F.linear(ln(sup, ln_g).float(), wg1.float()).to(dt)
    gate = torch.tanh((F.gelu(gh.float()) * w2.float().reshape(-1)).sum(-1, keepdim=True))
    out = sup.float() * gate + x.float()
    res = (out,) + ((ln(out, ln_f),) if ln_f is not None else ()) + ((gate.squeeze(-1),) if want_gate else ())
    return res if len(res) > 1 else out


def gcp_gate_residual(sup, h, w2, x, want_gate=False):
    gate = torch.tanh((F.gelu(h.float()) * w2.float()).sum(-1, keepdim=True))
    out = (sup.float() * gate + x.float()).to(x.dtype)
    return (out, gate.reshape(-1)) if want_gate else out


def _dcn_cols(x_nhwc, om, stride):
    B, H, W, C = x_nhwc.shape
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    n = Ho * Wo
    plane = om.shape[2] * om.shape[3]
    flat = om.reshape(B, -1)
    pos = torch.arange(n)
    ho, wo = (pos // Wo).float() * stride - 1, (pos % Wo).float() * stride - 1
    x = x_nhwc.float().reshape(B, H * W, C)
    cols = torch.zeros(B, n, 9, C)
    for k in range(9):
        dh, dw = flat[:, (2 * k) * n + pos], flat[:, (2 * k + 1) * n + pos]
        mk = torch.sigmoid(flat[:, 18 * plane + k * n + pos])
        hf, wf = ho[None] + k // 3 + dh, wo[None] + k % 3 + dw
        inside = (hf > -1) & (wf > -1) & (hf < H) & (wf < W)
        h0, w0 = torch.floor(hf), torch.floor(wf)
        lh, lw = hf - h0, wf - w0
        acc = torch.zeros(B, n, C)
        for dy, dx, wgt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            hh, ww = h0.long() + dy, w0.long() + dx
            ok = (hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1) & inside
            g = torch.gather(x, 1, (hh.clamp(0, H - 1) * W + ww.clamp(0, W - 1))[..., None].expand(-1, -1, C))
            acc += g * (wgt * ok.float())[..., None]
        cols[:, :, k] = acc * mk[..., None]
    return cols.reshape(B, n, 9 * C).to(x_nhwc.dtype), (Ho, Wo)


def align_scores(dot, tbias, tokidx, ctr, thr, want_cls=False, agg=0):
    v = (dot.float() + tbias[:, None, :]).clamp(-50000, 50000).sigmoid()
    L, MT = tokidx.shape[-2:]
    B = dot.shape[0]
    cls = torch.zeros(B, dot.shape[1], L)
    for b in range(B):
        tix = tokidx if tokidx.dim() == 2 else tokidx[b]
        for l in range(L):
            toks = [int(t) for t in tix[l] if int(t) >= 0]
            if toks:
                sel = v[b][:, toks]
                cls[b, :, l] = sel.mean(-1) if agg == 0 else (sel.max(-1)[0] if agg == 1 else sel.prod(-1) ** (1.0 / len(toks)))
    out = torch.where(cls > thr, (cls * ctr.float().sigmoid()[..., None]).clamp(min=1.17549435e-38), torch.full_like(cls, -1.0))
    return (out, cls) if want_cls else out


def align_fused(tok, tk, tbias, wbc, bbc, scales, tokidx, sizes, thr, agg=0, kv_max=0, want_cls=False, want_logits=False):
    """Plain-torch statement of mq_align_fused_fwd (per level: heads, logits, the align_scores emulation)."""
    B, N, C = tok.shape
    T = tk.shape[1]
    dots = torch.bmm(tok.float(), tk.float().transpose(1, 2))                                  # [B, N, T]
    bc = tok.float() @ wbc.float().t()[:, :8] + bbc[:8].float()                               # [B, N, 8]
    out = {"ranked": [], "reg": [], "ctr": bc[..., 4].contiguous()}
    cls_all = []
    off = 0
    for l, (h, w) in enumerate(sizes):
        hw = int(h) * int(w)
        d = dots[:, off:off + hw].contiguous()
        r = align_scores(d, tbias, tokidx, bc[:, off:off + hw, 4].contiguous(), thr, want_cls=True, agg=agg)      # centerness stays fp32
        out["ranked"].append(r[0])
        cls_all.append(r[1])
        out["reg"].append((bc[:, off:off + hw, :4] * scales[l].float()).contiguous())
        off += hw
    if want_cls:
        out["cls"] = cls_all
    if want_logits:
        out["logits"] = dots
    return out


def box_decode(val, flat, reg, anchors, label_ids, im_wh, boxes, scores, labels, HW, L, out_off):
    B, K = val.shape
    loc, l = flat // L, flat % L
    r = torch.gather(reg.float(), 1, loc[..., None].expand(-1, -1, 4))
    a = anchors[loc]
    w, h = a[..., 2] - a[..., 0] + 1, a[..., 3] - a[..., 1] + 1
    cx, cy = (a[..., 2] + a[..., 0]) / 2, (a[..., 3] + a[..., 1]) / 2
    lim = math.log(1000.0 / 16)
    pcx, pcy = r[..., 0] / 10 * w + cx, r[..., 1] / 10 * h + cy
    pw, ph = torch.exp((r[..., 2] / 5).clamp(max=lim)) * w, torch.exp((r[..., 3] / 5).clamp(max=lim)) * h
    W, H = im_wh[:, 0:1], im_wh[:, 1:2]
    bx = torch.stack([(pcx - 0.5 * (pw - 1)).clamp(min=0).minimum(W - 1), (pcy - 0.5 * (ph - 1)).clamp(min=0).minimum(H - 1),
                      (pcx + 0.5 * (pw - 1)).clamp(min=0).minimum(W - 1), (pcy + 0.5 * (ph - 1)).clamp(min=0).minimum(H - 1)], -1)
    ok = val > 0
    boxes[:, out_off:out_off + K] = torch.where(ok[..., None], bx, torch.zeros_like(bx))
    scores[:, out_off:out_off + K] = torch.where(ok, val.clamp(min=0).sqrt(), torch.full_like(val, -1.0))
    lab = label_ids[l] if label_ids.dim() == 1 else torch.gather(label_ids, 1, l)
    labels[:, out_off:out_off + K] = torch.where(ok, lab, torch.zeros_like(lab))


def ml_nms(boxes, labels, nvalid, thresh, max_keep=0, as_bool=True):
    from oracle.postprocess import ml_nms as ref
    B, N, _ = boxes.shape
    keep = torch.zeros(B, N, dtype=torch.bool)
    for b in range(B):
        nv = int(nvalid[b])
        if nv:
            sc = torch.arange(nv, 0, -1).float()          # already sorted by score
            keep[b, ref(boxes[b, :nv], sc, labels[b, :nv].float(), thresh)] = True
    return keep if as_bool else keep.to(torch.uint8)


def post_select(ranked, reg, anchors, ks, label_ids, im_wh):
    """mq_post_select_fwd restated with torch: per level the ks[l] largest positive values (ties: smaller flat index first), decoded."""
    B, _, L = ranked[0].shape
    tot = int(sum(ks))
    boxes, scores = torch.zeros(B, tot, 4), torch.full((B, tot), -1.0)
    labels, ids = torch.zeros(B, tot, dtype=torch.int32), torch.full((B, tot), 0x7FFFFFFF, dtype=torch.int32)
    off = idb = 0
    for r, g, a, k in zip(ranked, reg, anchors, ks):
        HW = r.shape[1]
        flatv = r.reshape(B, HW * L)
        val, flat = torch.sort(flatv, dim=1, descending=True, stable=True)          # stable: equal values keep index order
        val, flat = val[:, :k].contiguous(), flat[:, :k].contiguous()
        box_decode(val, flat, g, a, label_ids, im_wh, boxes, scores, labels, HW, L, off)
        ids[:, off:off + k] = torch.where(val > 0, (idb + flat).to(torch.int32), torch.full_like(flat, 0x7FFFFFFF).to(torch.int32))
        off += k
        idb += HW * L
    return boxes, scores, labels, ids


def post_select_supported(hws, ks, B, L):
    return len(hws) <= 8 and all(h * L < (1 << 22) and k <= 2048 for h, k in zip(hws, ks)) and sum(-(-h * L // 32768) for h in hws) <= 64


def post_sort(boxes, scores, labels, ks):
    """Merge of the per-level sorted lists: (score desc, level asc, position asc) = a STABLE descending sort of the concatenated lists."""
    key = torch.where(scores > 0, scores, torch.full_like(scores, -1.0))
    so, order = torch.sort(key, dim=1, descending=True, stable=True)
    live = so > 0
    bo = torch.where(live[..., None], torch.gather(boxes, 1, order[..., None].expand(-1, -1, 4)), torch.zeros_like(boxes))
    lo = torch.where(live, torch.gather(labels, 1, order), torch.zeros_like(labels))
    return bo, torch.where(live, so, torch.full_like(so, -1.0)), lo, live.sum(1).to(torch.int32)


def post_finalize(boxes, scores, labels, keep, K, K2):
    B, tot = scores.shape
    out = torch.zeros(B, K2, 6)
    out[..., 4] = -1.0
    counts = torch.zeros(B, dtype=torch.int32)
    for b in range(B):
        rows = [i for i in range(tot) if keep[b, i] and scores[b, i] > 0]
        take = rows[:K]
        if len(rows) >= K:
            ks_ = scores[b, rows[K - 1]]
            take += [i for i in rows[K:K2] if scores[b, i] == ks_]
        for j, i in enumerate(take):
            out[b, j, :4], out[b, j, 4], out[b, j, 5] = boxes[b, i], scores[b, i], float(labels[b, i])
        ovf = int(0 < K < K2 < tot and len(take) == K2 and float(scores[b, take[-1]]) == float(scores[b, take[K - 1]]))
        counts[b] = len(take) | (ovf << 16)
    return out, counts


def dyconv_branch_coef(y, Wsrc, gamma, beta, attn_w, attn_b, groups, eps, nbranches, wy=None, wx=None, sums=None):
    B, n, C = y.shape
    if sums is not None:                             # fused-statistics path: coefficients from (sum, sum sq, weighted sum)
        t = sums.sum(1)
        gs = t[..., 0].reshape(B, groups, C // groups).sum(-1)
        gss = t[..., 1].reshape(B, groups, C // groups).sum(-1)
        cnt = n * (C // groups)
        mean = gs / cnt
        var = (gss / cnt - mean * mean).clamp(min=0)
        rstd = (var + eps).rsqrt()
        sc = rstd.repeat_interleave(C // groups, 1) * gamma.float()
        sh = beta.float() - mean.repeat_interleave(C // groups, 1) * sc
        pooled = sc * t[..., 2] + sh
        a = F.relu6(F.relu(pooled @ attn_w + attn_b) + 3) / 6 / nbranches
        return torch.stack([a[:, None] * sc, a[:, None] * sh], -1)
    yf = y.float()
    g = yf.reshape(B, n, groups, C // groups)
    mean = g.mean((1, 3))
    var = g.var((1, 3), unbiased=False)
    rstd = (var + eps).rsqrt()
    mean_c = mean.repeat_interleave(C // groups, 1)
    rstd_c = rstd.repeat_interleave(C // groups, 1)
    sc = rstd_c * gamma.float()
    sh = beta.float() - mean_c * sc
    if wy is None:
        wmean = yf.mean(1)
    else:
        w = (wy[:, None] * wx[None, :]).reshape(-1)
        wmean = (yf * w[None, :, None]).sum(1)
    pooled = sc * wmean + sh
    a = F.relu6(F.relu(pooled @ attn_w + attn_b) + 3) / 6 / nbranches
    return torch.stack([a[:, None] * sc, a[:, None] * sh], -1)


def dyconv_fuse(branches, H, W, out=None):
    B, _, C = branches[0][0].shape
    acc = torch.zeros(B, H * W, C)
    for y, cf, hs, ws in branches:
        v = y.float().reshape(B, hs, ws, C).permute(0, 3, 1, 2)
        if (hs, ws) != (H, W):
            v = F.interpolate(v, size=(H, W), mode="bilinear", align_corners=True)
        v = v.permute(0, 2, 3, 1).reshape(B, H * W, C)
        acc = acc + v * cf[:, None, :, 0] + cf[:, None, :, 1]
    res = acc.to(branches[0][0].dtype)
    if out is not None:
        out.copy_(res)
        res = out
    return res, res.float().sum(1, keepdim=True)          # [B, nblk=1, C]


def dyrelu_coef(pool, n, w0, b0, w2, b2, out=None):
    """[B, 4, C] = (a1, b1, a2, b2) of  max(a1 x + b1, a2 x + b2)  (vldyhead.py:160-188: h_sigmoid of the two FCs, lambda_a = 2, init (1, 0))"""
    C = pool.shape[-1]
    y = pool.sum(1) / n
    y = F.relu6(F.linear(F.relu(F.linear(y, w0.float(), b0.float())), w2.float(), b2.float()) + 3) / 6
    a1, b1, a2, b2_ = torch.split(y, C, 1)
    coef = torch.stack([(a1 - 0.5) * 2 + 1, b1 - 0.5, (a2 - 0.5) * 2, b2_ - 0.5], 1)
    if out is not None:
        out.copy_(coef)
        return out
    return coef


def dyrelu_(x, pool, w0, b0, w2, b2):
    B, n, C = x.shape
    cf = dyrelu_coef(pool, n, w0, b0, w2, b2)
    xf = x.float()
    x.copy_(torch.max(xf * cf[:, 0, None] + cf[:, 1, None], xf * cf[:, 2, None] + cf[:, 3, None]).to(x.dtype))
    return x


def pool2x2_tokens(feats):
    return torch.cat([F.avg_pool2d(f.float(), 2).to(f.dtype).permute(0, 2, 3, 1).flatten(1, 2) for f in feats], 1)


def add_upsample_nearest_(dst, src):
    up = F.interpolate(src.permute(0, 3, 1, 2).float(), size=dst.shape[1:3], mode="nearest").permute(0, 2, 3, 1)
    dst.copy_((dst.float() + up).to(dst.dtype))
    return dst


def dyrelu_apply_(x, coef):
    xf = x.float()
    x.copy_(torch.max(xf * coef[:, 0, None] + coef[:, 1, None], xf * coef[:, 2, None] + coef[:, 3, None]).to(x.dtype))
    return x


def dyrelu_layer_norm(x, coef, sizes, gamma, beta, eps):
    """Plain-torch statement of mq_dyrelu_ln_fwd: per pyramid level DYReLU (consumed in fp32), then LayerNorm over the 256 channels."""
    out, off = torch.empty_like(x), 0
    for l, (h, w) in enumerate(sizes):
        xf = x[:, off:off + h * w].float()
        cf = coef[l]
        f = torch.max(xf * cf[:, 0, None] + cf[:, 1, None], xf * cf[:, 2, None] + cf[:, 3, None])
        out[:, off:off + h * w] = F.layer_norm(f, (x.shape[-1],), gamma.float(), beta.float(), eps).to(x.dtype)
        off += h * w
    return out


def conv3x3(x_nhwc, w_packed, bias, n_out, stride=1):
    B, H, W, C = x_nhwc.shape
    w = w_packed[:n_out].float().reshape(n_out, 3, 3, C).permute(0, 3, 1, 2)
    y = F.conv2d(x_nhwc.float().permute(0, 3, 1, 2), w, bias.float() if bias is not None else None, stride=stride, padding=1)
    return y.permute(0, 2, 3, 1).to(x_nhwc.dtype)


def conv3x3_nchw32(x_nhwc, w_packed, bias, n_out):
    return conv3x3(x_nhwc, w_packed, bias, n_out).permute(0, 3, 1, 2).float().contiguous()


def conv3x3_nchw32_group_supported(levels, n_out):
    return 0 < len(levels) <= 8 and n_out <= 32 and all(x.shape[3] == 256 for x in levels)


def conv3x3_nchw32_group(levels, w_packed, bias, n_out):
    return [conv3x3_nchw32(x, w_packed, bias, n_out) for x in levels]


def dcnv2(x_nhwc, om, w_packed, bias, stride, want_stats=False, wy=None, wx=None, mask_prob=False, tag=None, plain=False):
    from mq_det_amd import ops as _real
    if _real.dcn_is_tiled(w_packed):                 # the plan's LDS-tile-ordered copy (KERNELS["DCN_BDMA"]): back to rows for the torch form
        w_packed = _real.dcn_weight_rows(w_packed)
    cols, hw = _dcn_cols(x_nhwc.contiguous(), om, stride)
    y = F.linear(cols.float(), w_packed.float(), bias.float()).to(x_nhwc.dtype)
    if not want_stats:
        return y, hw
    yf = y.float()                                   # one "block" per image: [B, 1, C, 3] = (sum, sum sq, weighted sum)
    w = torch.full((yf.shape[1],), 1.0 / yf.shape[1]) if wy is None else (wy[:, None] * wx[None, :]).reshape(-1)
    sums = torch.stack([yf.sum(1), (yf * yf).sum(1), (yf * w[None, :, None]).sum(1)], -1)[:, None]
    return y, hw, sums


def layer_norm(x, gamma, beta, eps=1e-5, residual=None, want_sum=True, want_y32=False, want_y=True, clamp=0.0):
    """ops.layer_norm: y in the plan's activation dtype (gamma's), y32 fp32, sum fp32 when x or residual is fp32 (else the
    activation dtype, rounded BEFORE the statistics); clamp > 0: x clamped before the add, y / y32 after the affine."""
    act = gamma.dtype
    s = x.float()
    if clamp > 0:
        s = s.clamp(-clamp, clamp)
    sum_dt = torch.float32 if x.dtype == torch.float32 else act
    if residual is not None:
        s = s + residual.float()
        if residual.dtype == torch.float32:
            sum_dt = torch.float32
        if sum_dt != torch.float32:
            s = s.to(act).float()
    yf = F.layer_norm(s, (x.shape[-1],), gamma.float(), beta.float(), eps)
    out = []
    if want_y:
        out.append(yf.to(act).clamp(-clamp, clamp) if clamp > 0 else yf.to(act))
    if want_y32:
        out.append(yf.clamp(-clamp, clamp) if clamp > 0 else yf)
    if residual is not None and want_sum:
        out.append(s.to(sum_dt))
    return out[0] if len(out) == 1 else tuple(out)


def clamp_gelu_clamp(x, clamp):
    return F.gelu(x.float().clamp(-clamp, clamp)).to(x.dtype).clamp(-clamp, clamp)


def patch_merge_ln(x, gamma, beta, eps=1e-5):
    """mq_patch_merge_ln_fwd: Swin PatchMerging gather (swint.py:264-281: zero pad to even H / W, the four 2x2 phases concatenated
    along the channels in the order (0,0), (1,0), (0,1), (1,1)) + LayerNorm over 4C."""
    B, H, W, C = x.shape
    y = x.float()
    if H % 2 or W % 2:
        y = F.pad(y, (0, 0, 0, W % 2, 0, H % 2))
    y = torch.cat([y[:, 0::2, 0::2], y[:, 1::2, 0::2], y[:, 0::2, 1::2], y[:, 1::2, 1::2]], -1)
    y = F.layer_norm(y, (4 * C,), gamma.float(), beta.float(), eps)
    return y.reshape(B, -1, 4 * C).to(gamma.dtype)


def dcnv2_group(branches, want_stats=True, tag=None):
    return [dcnv2(br["x"], br["om"], br["w"], br["bias"], br["stride"], want_stats=True, wy=br.get("wy"), wx=br.get("wx"))
            for br in branches]


def vlfuse_i2t(v_ln, kf, vo, bias, out_bias, kv_len=None, max_kv=0, clamp=50000.0, variant=None):
    B, N, C = v_ln.shape
    T = kf.shape[2]
    s = torch.einsum("bnc,bhtc->bhnt", v_ln.float(), kf.float())
    masked = torch.zeros(B, kf.shape[1], T, dtype=torch.bool)
    if bias is not None:
        masked = bias < -1e29
        s = s + torch.where(masked, torch.zeros_like(bias), bias)[:, :, None, :]
    if clamp > 0:
        s = s.clamp(-clamp, clamp)
    if kv_len is not None:
        if max_kv > 0:
            assert int(kv_len.max()) <= max_kv
        masked = masked | (torch.arange(T)[None, None, :] >= kv_len.clamp(1, T)[:, None, None])
    s = s.masked_fill(masked[:, :, None, :].expand_as(s), -1e30)
    o = torch.einsum("bhnt,bhtc->bnc", s.softmax(-1), vo.float())
    return (v_ln.float() + out_bias.float() + o).to(v_ln.dtype)


def vlfuse_t2i(kf, v_ln, nsplit, clamp=50000.0, kv_len=None, key_mask=None, max_kv=0, variant=0):
    B, N, C = v_ln.shape
    T = kf.shape[2]
    s = torch.einsum("bhtc,bnc->bhtn", kf.float(), v_ln.float())
    if clamp > 0:
        s = s.clamp(-clamp, clamp)
    if key_mask is not None:
        assert key_mask.dtype == torch.uint8 and key_mask.shape[1] >= -(-N // 64) * 64 and key_mask.stride(0) % 4 == 0
        s = s.masked_fill(key_mask[:, None, None, :N].bool(), -1e30)
    o = torch.einsum("bhtn,bnc->bthc", s.softmax(-1), v_ln.float())
    if kv_len is not None:                      # 16-row blocks of pure padding come back as zeros
        dead = (torch.arange(T)[None, :] // 16) * 16 >= kv_len.clamp(1, T)[:, None]
        o = o.masked_fill(dead[:, :, None, None], 0.0)
    return o.reshape(B, T, kf.shape[1] * C).to(kf.dtype)


def roi_align(feat, rois, output_size, spatial_scale, sampling_ratio, aligned=True, reduce_mean=False):
    """Test-only stand-in: the oracle's ROIAlign on the same (strided) feature view."""
    from oracle import roi as oroi
    out = oroi.roi_align(feat.float().contiguous(), rois.float(), output_size, spatial_scale, sampling_ratio, aligned)
    return out.mean((-1, -2)) if reduce_mean else out


def swin_mlp2_unpack(w1f, w2f, C):
    """Inverse of ops.swin_mlp2_pack: fragment-major (w1f, w2f) -> (fc1.weight [4C, C], fc2.weight [C, 4C])."""
    from mq_det_amd.ops import swin_mlp_w2_perm, f32_operands, unsplit_planar_blocks
    if w1f.dtype == torch.float32 and f32_operands():           # the split-precise pack: blocks of [hi | lo] fp16 planes
        w1f, w2f = unsplit_planar_blocks(w1f), unsplit_planar_blocks(w2f)
    HID, KS, CT, NCH = 4 * C, C // 32, C // 16, 4 * C // 32
    w1 = w1f.reshape(NCH + 2, 2, KS, 4, 16, 8)[:NCH].permute(0, 1, 4, 2, 3, 5).reshape(HID, C)
    assert float(w1f.reshape(NCH + 2, -1)[NCH:].float().abs().max()) == 0.0          # the two zero chunks the pipeline reads ahead
    w2p = w2f.reshape(NCH, CT, 4, 16, 8).permute(1, 3, 0, 2, 4).reshape(C, HID)
    w2 = torch.empty_like(w2p)
    w2[:, swin_mlp_w2_perm(HID)] = w2p
    return w1.contiguous(), w2.contiguous()


def swin_mlp2(x, delta, ln_g, ln_b, eps, w1f, b1, w2f, b2, next_ln=None, flags=None):
    """Plain-torch statement of mq_swin_mlp2_fwd (un-packs the fragment-major weights first)."""
    C = x.shape[-1]
    w1, w2 = swin_mlp2_unpack(w1f, w2f, C)
    act = ln_g.dtype
    xp = x.float() + (delta.float() if delta is not None else 0.0)
    h = F.layer_norm(xp, (C,), ln_g.float(), ln_b.float(), eps).to(act)
    hid = F.gelu(F.linear(h.float(), w1.float(), b1.float())).to(act)
    out = xp + F.linear(hid.float(), w2.float(), b2.float())
    if next_ln is None:
        return out
    ng, nb, ne = next_ln
    return out, F.layer_norm(out, (C,), ng.float(), nb.float(), ne).to(act)


def ms_deform_attn(value, spatial_shapes, sampling_locations, attention_weights, out_dtype=None):
    from oracle.gdino import ms_deform_attn_core
    out = ms_deform_attn_core(value.float(), spatial_shapes, sampling_locations.float(), attention_weights.float())
    return out.to(out_dtype or value.dtype)


def image_key_mask(mask):
    """Pure tensor code in the product (no kernel): restated so that patching it over itself cannot recurse."""
    B, N = mask.shape
    out = torch.ones(B, -(-N // 64) * 64, dtype=torch.uint8, device=mask.device)
    out[:, :N] = mask.to(torch.uint8)
    return out


def ms_deform_attn_q(value, spatial_shapes, qproj, ref, heads, out_dtype=None, valid_hw=None):
    """Plain-torch statement of mq_msdeform_attn_q_fwd: softmax + sampling locations, then the unfused emulation."""
    B, S, C = value.shape
    if valid_hw is not None:                                 # rows / columns outside the valid rectangle read as zero
        value, s0 = value.clone(), 0
        for l, (h, w) in enumerate(spatial_shapes):
            ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
            for b in range(B):
                bad = (ys >= int(valid_hw[b, l, 0])) | (xs >= int(valid_hw[b, l, 1]))
                value[b, s0:s0 + h * w][bad.reshape(-1)] = 0
            s0 += h * w
    Q, L, P = qproj.shape[1], len(spatial_shapes), 4
    n = heads * L * P
    qp = qproj.float()
    off = qp[..., :2 * n].reshape(B, Q, heads, L, P, 2)
    aw = qp[..., 2 * n:].reshape(B, Q, heads, L * P).softmax(-1).reshape(B, Q, heads, L, P)
    if ref.shape[-1] == 2:
        norm = torch.tensor([[w, h] for h, w in spatial_shapes], dtype=torch.float32)
        loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
    return ms_deform_attn(value.reshape(B, S, heads, C // heads).contiguous(), spatial_shapes, loc.contiguous(), aw.contiguous(),
                          out_dtype)


def dyconv_epilogue_group(levels, w0, b0, w2, b2, relu_coef):
    for l, (branches, H, W, out) in enumerate(levels):
        _, pool = dyconv_fuse(branches, H, W, out=out)
        dyrelu_coef(pool, H * W, w0, b0, w2, b2, out=relu_coef[l])
    return relu_coef


def dyconv_coef_group(items, attn_w, attn_b, groups, eps):
    out = []
    for it in items:
        B, _, C, _ = it["sums"].shape
        y_shape = torch.empty(B, int(it["n"]), C, device="meta")          # only its shape is read on the fused-statistics path
        out.append(dyconv_branch_coef(y_shape, 0, it["gamma"], it["beta"], attn_w, attn_b, groups, eps, it["nbranches"], sums=it["sums"]))
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# every emulated entry point, in one place: tests patch them into mq_det_amd.ops (or into a stand-in namespace) with these helpers
NAMES = ("attention", "attention4", "attention_text", "bert_attention_qkv", "patch_embed", "window_attention", "window_attention_qkv", "gcp_sparse_attention", "gcp_gate_residual", "gcp_attention", "dcnv2_group", "align_scores", "align_fused",
         "dyconv_branch_coef", "dyconv_coef_group", "dyconv_epilogue_group", "dyconv_fuse", "dyrelu_", "dyrelu_coef", "dyrelu_apply_", "dyrelu_layer_norm", "add_upsample_nearest_", "pool2x2_tokens", "conv3x3", "conv3x3_nchw32", "conv3x3_nchw32_group", "conv3x3_nchw32_group_supported", "dcnv2", "layer_norm", "clamp_gelu_clamp", "vlfuse_i2t",
         "vlfuse_t2i", "box_decode", "ml_nms", "post_select", "post_select_supported", "post_sort", "post_finalize", "roi_align", "swin_mlp2", "patch_merge_ln", "ms_deform_attn", "ms_deform_attn_q", "image_key_mask")


def patch_into(monkeypatch, ops_module):
    """monkeypatch every emulated entry point into the REAL mq_det_amd.ops module (wrappers that need a GPU are replaced)."""
    g = globals()
    for n in NAMES:
        monkeypatch.setattr(ops_module, n, g[n])


def namespace(real_ops):
    """A stand-in for the `ops` module of pipeline.py / gdino_pipeline.py: emulated entry points + the real module's host-side
    helpers and tables (kernel selection defaults, pad sizes, weight permutations)."""
    import types
    g = globals()
    fake = types.SimpleNamespace(**{n: g[n] for n in NAMES})
    for n in ("patch_embed_pack", "SWIN_MLP_WIDTHS", "WINDOW_QKV_WIDTHS", "window_qkv_fused", "SCORE_AGG", "window_pad", "pad_rel_bias", "swin_mlp_w2_perm", "swin_mlp2_pack", "timing_active", "attention_text_fits", "bert_attention_qkv_fits", "gcp_attention_fits", "f32_operands", "pack_b_fragments", "unpack_b_fragments", "dcn_bdma", "dcn_weight_tiles", "pool2x2_tokens_supported"):
        setattr(fake, n, getattr(real_ops, n))
    fake.KERNELS = dict(real_ops.KERNEL_DEFAULTS)       # the default kernel selection: the glue of the promoted variants is what runs
    return fake
