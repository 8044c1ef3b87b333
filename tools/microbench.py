#!/usr/bin/env python
"""Stand-alone timings (HIP events, 20 launches after 3 warm-ups) of kernels that the configs[1] bench does not exercise or
that are worth seeing in isolation:  python tools/microbench.py [out.json]
  * mq_msdeform_attn_fwd at the MQ-GroundingDINO encoder shape (BASELINE configs[4]: B = 16, 4 levels of 800x1344, Q = 22 323)
  * mq_swin_mlp2_fwd per Swin stage at B = 8
  * mq_window_attn_fwd with 144-token windows (Swin-L stage 1 at B = 4)
Rates are ALGORITHMIC bytes / flops over the event-measured launch time."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mq_det_amd import ops  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    dev = torch.device("cuda:0")
    ops.load_library()
    g = torch.Generator().manual_seed(0)
    out = []
    only = os.environ.get("MQ_MICRO_ONLY", "")
    if only in ("", "msda"):
        msda(dev, g, out)
    if only in ("", "swin"):
        swin(dev, g, out)
    if only in ("", "align"):
        align(dev, g, out)
    if only in ("", "window"):
        window(dev, g, out)
    if only in ("", "window_qkv"):
        window_qkv(dev, g, out)
    if only in ("", "dyconv"):
        dyconv_parts(dev, g, out)
    if only in ("", "dcn"):
        dcn(dev, g, out)
    if only in ("", "vlfuse"):
        vlfuse(dev, g, out)
        vlfuse_text(dev, g, out)
    if only in ("", "bert_attn"):
        bert_attn(dev, g, out)
    if only in ("", "gcp_attn"):
        gcp_attn(dev, g, out)
    if only == "t2i_sweep":
        t2i_sweep(dev, g, out)
    if only == "offset_conv":
        offset_conv(dev, g, out)
    for r in out:
        print(json.dumps(r))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


def gcp_attn(dev, g, out):
    """The attention half of a GCP block on the compacted text (T = 144): ONE launch of mq_gcp_attn_fwd against the eight launches it replaces
    (LayerNorm, to_q GEMM, mq_gcp_sparse_attn_fwd, to_out GEMM, LayerNorm, gate GEMM, mq_gcp_gate_residual_fwd, LayerNorm)."""
    import torch.nn.functional as F
    T, V, S = 144, 200, 5
    h = lambda *s_, sc=1.0: (torch.randn(*s_, generator=g) * sc).half().to(dev)          # noqa: E731
    wq, wout, wg1, w2 = h(512, 768, sc=768 ** -0.5), h(768, 512, sc=512 ** -0.5), h(384, 768, sc=768 ** -0.5), h(384, sc=0.1)
    lns = [(h(768, sc=0.1) + 1, h(768, sc=0.1)) for _ in range(3)]
    pq, pout, pg1 = (ops.pack_b_fragments(w_) for w_ in (wq, wout, wg1))      # packed once, as the pipeline does
    for B in (8, 64):
        x = torch.randn(B, T, 768, generator=g).to(dev)
        kv = h(B, V, 1024)
        idx = torch.randint(0, V, (B, T, S), generator=g).to(torch.int32).to(dev)

        def unfused():
            q = F.linear(ops.layer_norm(x, *lns[0], 1e-5), wq)
            sup = F.linear(ops.gcp_sparse_attention(q, kv, idx), wout)
            gh = F.linear(ops.layer_norm(sup, *lns[1], 1e-5), wg1)
            xo = ops.gcp_gate_residual(sup, gh, w2, x)
            return xo, ops.layer_norm(xo, *lns[2], 1e-5)
        fl = 2.0 * B * T * (768 * 512 + 512 * 768 + 768 * 384)
        rec = {"kernel": "gcp attention half", "B": B, "T": T, "algorithmic_gflop": round(fl / 1e9, 2), "unfused_8_launches_ms": round(timeit(unfused), 4)}
        for rb in (16, 32):
            ms = timeit(lambda: ops.gcp_attention(x, kv, idx, pq, pout, pg1, w2, lns[0], lns[1], lns[2], rows_per_block=rb, packed=True))
            rec[f"fused_rb{rb}_ms"] = round(ms, 4)
            rec[f"fused_rb{rb}_tflops"] = round(fl / ms / 1e9, 1)
        # what the rb16 time is made of: the same launch without the weight stream (1), the key / value gather (2), the erf / tanh arithmetic (4),
        # the MFMAs (8), all four (15) -- rows_per_block = 16 | mask << 8, outputs meaningless
        for mask in (1, 2, 4, 8, 15):
            ms = timeit(lambda: ops.gcp_attention(x, kv, idx, pq, pout, pg1, w2, lns[0], lns[1], lns[2], rows_per_block=16 | mask << 8, packed=True))
            rec[f"fused_rb16_without_{mask}_ms"] = round(ms, 4)
        out.append(rec)


def bert_attn(dev, g, out):
    """The attention half of a BERT layer (C = 768, 12 heads) on the compacted text of the 141-token benchmark caption (T = 144): ONE launch of
    mq_bert_attn_qkv_fwd against the round-4 path (library qkv GEMM + mq_attn_text_fwd), at the headline batch and at the north-star's B = 64;
    the round-4 path on the uncompacted T = 256 rows beside it."""
    import torch.nn.functional as F
    C, H = 768, 12
    w = (torch.randn(3 * C, C, generator=g) / C ** 0.5).half().to(dev)
    bq = (torch.randn(3 * C, generator=g) * 0.1).half().to(dev)
    wp = ops.pack_b_fragments(w)                                             # packed once, as the pipeline does
    for B in (8, 64):
        for T, kv in ((144, 141), (256, 141)):
            x = torch.randn(B, T, C, generator=g).half().to(dev)
            kl = torch.full((B,), kv, dtype=torch.int32, device=dev)
            kb = torch.zeros(B, T, device=dev)
            kb[:, kv:] = -1e30
            k16 = -(-kv // 16) * 16
            fl = 2.0 * B * T * C * 3 * C + 4.0 * B * H * T * k16 * 64
            ms_f = timeit(lambda: ops.bert_attention_qkv(x, wp, bq, H, key_bias=kb, kv_len=kl, packed=True))
            ms_g = timeit(lambda: F.linear(x, w, bq))
            qkv = F.linear(x, w, bq)
            ms_a = timeit(lambda: ops.attention_text(qkv, H, key_bias=kb, kv_len=kl, max_kv=kv))
            out.append({"kernel": "bert attention half", "B": B, "T": T, "live_tokens": kv, "algorithmic_gflop": round(fl / 1e9, 2),
                        "fused_ms": round(ms_f, 4), "fused_tflops": round(fl / ms_f / 1e9, 1), "fused_frac_of_mfma_peak": round(fl / ms_f / 1e9 / 2500.0, 4),
                        "qkv_gemm_ms": round(ms_g, 4), "attn_text_ms": round(ms_a, 4), "unfused_ms": round(ms_g + ms_a, 4)})


def msda(dev, g, out):
    # ---- MSDeformAttn, encoder self-attention shape
    shapes = [(100, 168), (50, 84), (25, 42), (13, 21)]
    S = sum(h * w for h, w in shapes)
    B = 16
    v = torch.randn(B, S, 8, 32, generator=g).half().to(dev)
    ref = torch.rand(B, S, 1, 1, 1, 2, generator=g)
    loc = (ref + torch.randn(B, S, 8, 4, 4, 2, generator=g) * 0.02).clamp(0, 1).to(dev).contiguous()     # offsets of a few pixels
    attn = torch.rand(B, S, 8, 16, generator=g).softmax(-1).reshape(B, S, 8, 4, 4).to(dev).contiguous()
    ms = timeit(lambda: ops.ms_deform_attn(v, shapes, loc, attn))
    nb = v.numel() * 2 + loc.numel() * 4 + attn.numel() * 4 + B * S * 256 * 2
    gathered = B * S * 8 * 16 * 4 * 64                     # bytes the gather touches (4 corners x 64 B per sample), mostly L2 hits
    out.append({"kernel": "msda_kernel (encoder, B=16, Q=S=22323, 8 heads x 32, 4 levels x 4 points, fp16 values)", "ms": round(ms, 3),
                "algorithmic_GBs": round(nb / ms / 1e6, 1), "gather_GBs": round(gathered / ms / 1e6, 1),
                "algorithmic_bytes": nb, "frac_of_hbm_peak": round(nb / ms / 1e6 / 8000, 3)})


def swin(dev, g, out):
    # ---- fused Swin MLP per stage (B = 8, 800x1344): mq_swin_mlp2_fwd: erf / table GELU, with and without the pass / tail split; one exact pass (256 x 128 tokens) at C = 384
    for C, M in ((96, 8 * 67200), (192, 8 * 16800), (384, 8 * 4200), (384, 256 * 128), (384, 52 * 16)):
        x = torch.randn(M, C, generator=g).to(dev)
        d = torch.randn(M, C, generator=g).half().to(dev)
        lg, lb = torch.ones(C).half().to(dev), torch.zeros(C).half().to(dev)
        w1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).half()
        b1 = (torch.randn(4 * C, generator=g) * 0.1).half().to(dev)
        w2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).half()
        b2 = torch.zeros(C).half().to(dev)
        w1f, w2f = (t.to(dev) for t in ops.swin_mlp2_pack(w1, w2))
        fl, nb = 16.0 * M * C * C, M * C * (4 + 2 + 4 + 2)
        runs = []
        for flags in (0, 1, 2, 3, 4):
            if flags & 4 and M > 40000:
                continue
            runs.append((f"v2[{'tail only' if flags & 4 else 'unsplit' if flags & 1 else 'split'},{'table' if flags & 2 else 'erf'}]",
                         lambda flags=flags: ops.swin_mlp2(x, d, lg, lb, 1e-5, w1f, b1, w2f, b2, next_ln=(lg, lb, 1e-5), flags=flags)))
        ref = None
        for name, fn in runs:
            ms = timeit(fn)
            o = fn()[0].float()
            ref = o if ref is None else ref
            out.append({"kernel": f"swin_mlp {name} C={C} M={M}", "ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1),
                        "frac_of_mfma_peak": round(fl / ms / 1e9 / 2500, 3), "algorithmic_GBs": round(nb / ms / 1e6, 1),
                        "max_abs_diff_vs_first_variant": round(float((o - ref).abs().max()), 6)})


def align(dev, g, out):
    # ---- heads + alignment + scoring at the bench shape (B = 8, N = 22400, 141 live text tokens, 40 labels)
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    B, N, T, L, MT = 8, sum(h * w for h, w in sizes), 256, 40, 4
    tok = (torch.randn(B, N, 256, generator=g) * 0.5).half().to(dev)
    tk = (torch.randn(B, T, 256, generator=g) * 0.1).half().to(dev)
    tbias = (torch.randn(B, T, generator=g) * 0.3 - 2.0).to(dev)
    wbc = torch.zeros(16, 256)
    wbc[:5] = torch.randn(5, 256, generator=g) * 0.05
    wbc = wbc.half().to(dev)
    bbc, scales = torch.zeros(8).to(dev), torch.ones(5).to(dev)
    tokidx = torch.randint(0, 141, (L, MT), generator=g).to(torch.int32).to(dev)
    for kv in (141, 0):
        ms = timeit(lambda: ops.align_fused(tok, tk, tbias, wbc, bbc, scales, tokidx, sizes, 0.05, kv_max=kv))
        nb = tok.numel() * 2 + B * N * L * 4 + B * N * 8 + B * N * 4
        out.append({"kernel": f"align_fused B={B} N={N} T={T} live={kv or T} L={L}", "ms": round(ms, 4), "algorithmic_GBs": round(nb / ms / 1e6, 1),
                    "frac_of_hbm_peak": round(nb / ms / 1e6 / 8000, 3)})
    # the round-2 path it replaces: bmm (fp16 logits to HBM) + box / centerness GEMM + 5 x align_scores
    w8 = wbc[:8].contiguous()

    def old():
        dots = torch.bmm(tok, tk.transpose(1, 2))
        bc = torch.nn.functional.linear(tok, w8)
        off = 0
        for (h, w) in sizes:
            ops.align_scores(dots[:, off:off + h * w], tbias, tokidx, bc[:, off:off + h * w, 4].contiguous(), 0.05)
            off += h * w
    out.append({"kernel": "round-2 path: bmm + head GEMM + 5 x align_scores (same shape)", "ms": round(timeit(old), 4)})


def vlfuse(dev, g, out):
    # ---- VLFuse image side at the bench shape (B = 8, N = 22 400 image tokens, 8 heads x 256, T = 256 with 141 / 81 / 256 live keys):
    # Q tile in LDS against Q fragments in registers (129 .. 160 keys), and the ablation variants.  flops = 4 * B * heads * N * keys_visited * 256
    B, N, T = 8, 22400, 256
    v = torch.randn(B, N, 256, generator=g).half().to(dev)
    kf = (torch.randn(B, 8, T, 256, generator=g) / 8).half().to(dev)
    vo = torch.randn(B, 8, T, 256, generator=g).half().to(dev)
    bias = torch.randn(B, 8, T, generator=g).to(dev)
    ob = torch.zeros(256).half().to(dev)
    for live in (141, 81, 256):
        kv = torch.full((B,), live, dtype=torch.int32, device=dev)
        ref = None
        # 101 ... 115: the first kernel WITHOUT global tile loads (bit 0) / LDS commits (1) / softmax (2) / fragment reads + MFMAs (3)
        for variant in (1, 0) + ((101, 103, 104, 108, 111, 115) if live == 141 else ()):
            fn = lambda variant=variant: ops.vlfuse_i2t(v, kf, vo, bias, ob, kv_len=kv, max_kv=live, variant=variant)  # noqa: E731
            ms = timeit(fn)
            o = fn().float()
            ref = o if ref is None else ref
            visited = -(-live // 16) * 16
            fl = 4.0 * B * 8 * N * visited * 256
            nb = v.numel() * 2 * 2 + kf.numel() * 2 * 2
            name = {1: "Q tile in LDS beyond 128 keys", 0: "default (Q in registers for 129 .. 160 keys)"}.get(variant, f"first kernel, ablation bits {variant - 100:04b} (mfma|softmax|commits|loads removed)")
            out.append({"kernel": f"vlfuse_i2t {name} B={B} N={N} live keys={live}", "ms": round(ms, 4),
                        "TFLOPs": round(fl / ms / 1e9, 1), "frac_of_mfma_peak": round(fl / ms / 1e9 / 2500, 3),
                        "algorithmic_GBs": round(nb / ms / 1e6, 1), "max_abs_diff_vs_first": round(float((o - ref).abs().max()), 6)})


def offset_conv(dev, g, out):
    # ---- the offset conv of ONE DyConv layer at the bench pyramid (B = 8, five levels as slices of one token buffer): the per-level
    # kernel on one stream and forked over five streams (what dyconv_tokens does), against the grouped launch (OFFSET_CONV_VARIANT 3)
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    B, C = 8, 256
    w = torch.zeros(32, 9 * C)
    w[:27] = torch.randn(27, 9 * C, generator=g) / 48
    w, bias = w.half().to(dev), torch.randn(27, generator=g).half().to(dev)
    tok = torch.randn(B, sum(h * w_ for h, w_ in sizes), C, generator=g).half().to(dev)
    lv, off = [], 0
    for (h, w_) in sizes:
        lv.append(tok[:, off:off + h * w_].reshape(B, h, w_, C))
        off += h * w_
    nb = tok.numel() * 2 + B * 27 * 4 * sum(h * w_ for h, w_ in sizes)
    side = [torch.cuda.Stream() for _ in range(4)]

    def per_level():
        return [ops.conv3x3_nchw32(x, w, bias, 27) for x in lv]

    def forked():
        main = torch.cuda.current_stream()
        for s_ in side:
            s_.wait_stream(main)
        res = [None] * 5
        for l in range(1, 5):
            with torch.cuda.stream(side[l - 1]):
                res[l] = ops.conv3x3_nchw32(lv[l], w, bias, 27)
        res[0] = ops.conv3x3_nchw32(lv[0], w, bias, 27)
        for s_ in side:
            main.wait_stream(s_)
        return res

    ref = per_level()
    got = ops.conv3x3_nchw32_group(lv, w, bias, 27)
    diff = max(float((a - b_).abs().max()) for a, b_ in zip(ref, got))
    for name, fn in (("per-level kernel v2, one stream (5 launches)", per_level), ("per-level kernel v2, five streams (5 launches + fork / join)", forked),
                     ("grouped launch v3 (1 launch, persistent workgroups, weights in registers)", lambda: ops.conv3x3_nchw32_group(lv, w, bias, 27))):
        ms = timeit(fn, n=50, warm=5)
        out.append({"kernel": f"offset conv of one DyConv layer, B={B}, 5 levels: {name}", "ms": round(ms, 4), "algorithmic_GBs": round(nb / ms / 1e6, 1),
                    "frac_of_hbm_peak": round(nb / ms / 1e6 / 8000, 3), "max_abs_diff_grouped_vs_per_level": diff})
    for l, x in enumerate(lv[:2]):
        ms = timeit(lambda: ops.conv3x3_nchw32_group([x], w, bias, 27), n=50, warm=5)
        ms2 = timeit(lambda: ops.conv3x3_nchw32(x, w, bias, 27), n=50, warm=5)
        out.append({"kernel": f"offset conv, level {l} alone {sizes[l]}: grouped kernel / per-level kernel", "ms": round(ms, 4), "ms_per_level_kernel": round(ms2, 4)})


def dyconv_parts(dev, g, out):
    # ---- DyConv side kernels at the bench shape (B = 8): the offset conv per pyramid level (variants 2 / 3), DYReLU coefficients,
    # and LayerNorm(DYReLU(x)) fused against the two passes it replaces
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    B, C = 8, 256
    w = torch.zeros(32, 9 * C)
    w[:27] = torch.randn(27, 9 * C, generator=g) / 48
    w, bias = w.half().to(dev), torch.randn(27, generator=g).half().to(dev)
    for (H, W) in sizes[:3]:
        x = torch.randn(B, H, W, C, generator=g).half().to(dev)
        for variant in (1, 2):
            ops.KERNELS["OFFSET_CONV_VARIANT"] = variant
            ms = timeit(lambda: ops.conv3x3_nchw32(x, w, bias, 27))
            nb = x.numel() * 2 + B * 27 * H * W * 4
            out.append({"kernel": f"offset conv v{variant} B={B} {H}x{W}x{C} -> 27", "ms": round(ms, 4), "algorithmic_GBs": round(nb / ms / 1e6, 1),
                        "frac_of_hbm_peak": round(nb / ms / 1e6 / 8000, 3)})
    ops.KERNELS["OFFSET_CONV_VARIANT"] = ops.KERNEL_DEFAULTS["OFFSET_CONV_VARIANT"]
    N = sum(h * w_ for h, w_ in sizes)
    tok = torch.randn(B, N, C, generator=g).half().to(dev)
    gam, bet = torch.ones(C).half().to(dev), torch.zeros(C).half().to(dev)
    coef = torch.randn(len(sizes), B, 4, C, generator=g).to(dev)
    ms = timeit(lambda: ops.dyrelu_layer_norm(tok, coef, sizes, gam, bet, 1e-5))
    out.append({"kernel": f"LayerNorm(DYReLU(x)) fused, [B={B}, N={N}, 256]", "ms": round(ms, 4), "algorithmic_GBs": round(2 * tok.numel() * 2 / ms / 1e6, 1)})

    def two_pass():
        off = 0
        for l, (h, w_) in enumerate(sizes):
            ops.dyrelu_apply_(tok[:, off:off + h * w_], coef[l])
            off += h * w_
        return ops.layer_norm(tok, gam, bet, 1e-5)
    ms = timeit(two_pass)
    out.append({"kernel": "5 x dyrelu_apply + LayerNorm (the passes it replaces, one stream)", "ms": round(ms, 4)})
    w0, b0 = (torch.randn(64, C, generator=g) / 16).half().to(dev), torch.zeros(64).half().to(dev)
    w2, b2 = (torch.randn(4 * C, 64, generator=g) / 8).half().to(dev), torch.zeros(4 * C).half().to(dev)
    for (H, W) in sizes[:2]:
        pool = torch.randn(B, (H * W + 127) // 128, C, generator=g).to(dev)
        ms = timeit(lambda: ops.dyrelu_coef(pool, H * W, w0, b0, w2, b2))
        out.append({"kernel": f"dyrelu_coef {H}x{W} ({pool.shape[1]} partials per image)", "ms": round(ms, 4)})


def dcn(dev, g, out):
    # ---- the grouped DCNv2 launch of one DyConv layer at the bench shape (B = 8, five levels, 13 branches) and its ablation variants
    # (the kernel without its gather loads / blend / weight-tile loads / fragment reads + MFMAs: results garbage, timing only)
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    B, C = 8, 256
    lv = [torch.randn(B, h, w, C, generator=g).half().to(dev) for h, w in sizes]
    om = [(torch.randn(B, 27, h, w, generator=g) * 1.5).to(dev) for h, w in sizes]
    wts = [(torch.randn(256, 9 * C, generator=g) / 48).half().to(dev) for _ in range(3)]
    bias = torch.zeros(256).half().to(dev)
    branches = []
    for l in range(5):
        spec = [(1, lv[l], 1)] + ([(2, lv[l - 1], 2)] if l > 0 else []) + ([(0, lv[l + 1], 1)] if l < 4 else [])
        for k, x, stride in spec:
            branches.append({"x": x, "om": om[l], "w": wts[k], "bias": bias, "stride": stride, "wy": None, "wx": None})
    fl = sum(2.0 * B * (((b_["x"].shape[1] - 1) // b_["stride"] + 1) * ((b_["x"].shape[2] - 1) // b_["stride"] + 1)) * 2304 * 256 for b_ in branches)
    names = {0: "full", 1: "no gather loads", 2: "no blend", 3: "no gather loads, no blend", 4: "no weight-tile loads", 7: "no loads, no blend",
             8: "no fragment reads / MFMAs", 15: "skeleton (barriers, sampling state, LDS stores, epilogue)"}
    abls = tuple(int(a) for a in os.environ.get("MQ_DCN_ABL_LIST", "0,1,2,3,4,7,8,15").split(","))
    for abl in abls:
        ms = timeit(lambda: ops.dcnv2_group(branches, want_stats=True, ablation=abl))
        out.append({"kernel": f"dcn_igemm8_kernel<16>, 13 branches, B={B}: {names[abl]}", "ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1),
                    "frac_of_mfma_peak": round(fl / ms / 1e9 / 2500, 3)})


def vlfuse_text(dev, g, out):
    # ---- VLFuse text side at the bench shape: live (head, row block) units packed (max_kv given) vs one grid row per 16 text rows of T
    from mq_det_amd.modeling.pipeline import _nsplit, _nsplit_t2i
    B, N, T = 8, 22400, 256
    v = torch.randn(B, N, 256, generator=g).half().to(dev)
    kf = (torch.randn(B, 8, T, 256, generator=g) / 8).half().to(dev)
    for live in (141, 81, 256):
        kv = torch.full((B,), live, dtype=torch.int32, device=dev)
        tiles = -(-N // 64)
        for name, ns, mk in (("packed, nsplit by passes x tiles", _nsplit_t2i(B, 8, live, tiles), live), ("packed, nsplit 6", 6, live),
                             ("grid sized for all T rows (dead workgroups exit early), nsplit 6", 6, 0)):
            ms = timeit(lambda: ops.vlfuse_t2i(kf, v, ns, kv_len=kv, max_kv=mk))
            fl = 4.0 * B * 8 * (-(-live // 16) * 16) * N * 256
            out.append({"kernel": f"vlfuse_t2i + combine, {name} (nsplit={ns}) B={B} N={N} live rows={live}", "ms": round(ms, 4),
                        "TFLOPs": round(fl / ms / 1e9, 1), "frac_of_mfma_peak": round(fl / ms / 1e9 / 2500, 3)})
        if live == 141:
            ns = _nsplit_t2i(B, 8, live, tiles)
            for abl in (1, 3, 4, 8, 11, 15):
                ms = timeit(lambda: ops.vlfuse_t2i(kf, v, ns, kv_len=kv, max_kv=live, variant=100 + abl))
                out.append({"kernel": f"vlfuse_t2i + combine, ablation bits {abl:04b} (mfma|softmax|commits|loads removed) nsplit={ns} live rows={live}", "ms": round(ms, 4)})


def t2i_sweep(dev, g, out):
    # ---- VLFuse text side: key split sweep at B = 4 (BASELINE configs[3]) and B = 8 -- what does the cost model of _nsplit_t2i miss?
    from mq_det_amd.modeling.pipeline import _nsplit_t2i
    N, T, live = 22400, 256, 141
    for B in (4, 8):
        v = torch.randn(B, N, 256, generator=g).half().to(dev)
        kf = (torch.randn(B, 8, T, 256, generator=g) / 8).half().to(dev)
        kv = torch.full((B,), live, dtype=torch.int32, device=dev)
        chosen = _nsplit_t2i(B, 8, live, -(-N // 64))
        for ns in sorted({chosen, 2, 3, 4, 5, 7, 10, 14, 20, 28}):
            ms = timeit(lambda: ops.vlfuse_t2i(kf, v, ns, kv_len=kv, max_kv=live))
            out.append({"kernel": f"vlfuse_t2i + combine B={B} nsplit={ns}{' (chosen)' if ns == chosen else ''}", "ms": round(ms, 4),
                        "workgroups": B * ns * 9})


def window_qkv(dev, g, out):
    # ---- Swin stages 1 / 2 (C = 96 / 192, B = 8): qkv GEMM + window attention against the kernel with the projection inside
    import torch.nn.functional as F
    for (H, W, C, heads) in ((200, 336, 96, 3), (100, 168, 192, 6)):
        _window_qkv_one(dev, g, out, F, 8, H, W, C, heads, 7)


def _window_qkv_one(dev, g, out, F, B, H, W, C, heads, ws):
    x = torch.randn(B, H, W, C, generator=g).half().to(dev)
    w = (torch.randn(3 * C, C, generator=g) / C ** 0.5).half().to(dev)
    bias = (torch.randn(3 * C, generator=g) * 0.1).half().to(dev)
    rel = ops.pad_rel_bias((torch.randn(heads, ws * ws, ws * ws, generator=g) * 0.1).to(dev), ws)
    for shift in (0, 3):
        two = lambda: ops.window_attention(F.linear(x, w, bias), bias, rel, heads, ws, shift)  # noqa: E731
        one = lambda: ops.window_attention_qkv(x, w, bias, rel, heads, ws, shift)              # noqa: E731
        t2, t1 = timeit(two), timeit(one)
        d = float((one().float() - two().float()).abs().max())
        nb = 2 * x.numel() * 2
        out.append({"kernel": f"Swin C={C} attention, shift={shift}: qkv GEMM + window_attn", "ms": round(t2, 4), "hbm_bytes_incl_qkv_tensor": 2 * x.numel() * 2 + 2 * 3 * x.numel() * 2})
        out.append({"kernel": f"Swin C={C} attention, shift={shift}: window_attn_qkv (projection inside)", "ms": round(t1, 4), "algorithmic_GBs": round(nb / t1 / 1e6, 1),
                    "TFLOPs": round((2.0 * B * H * W * C * 3 * C + 4.0 * B * H * W * 49 * C) / t1 / 1e9, 1), "max_abs_diff": round(d, 5)})


def window(dev, g, out):
    # ---- window attention, 144-token windows (Swin-L stage 1, B = 4)
    Bn, H, W, C, heads, ws = 4, 200, 336, 192, 6, 12
    qkv = torch.randn(Bn, H, W, 3 * C, generator=g).half().to(dev)
    qb = torch.zeros(3 * C).half().to(dev)
    rel = ops.pad_rel_bias(torch.randn(heads, ws * ws, ws * ws, generator=g), ws).to(dev)
    for shift in (0, ws // 2):
        ms = timeit(lambda: ops.window_attention(qkv, qb, rel, heads, ws, shift))
        nb = qkv.numel() * 2 + Bn * H * W * C * 2
        out.append({"kernel": f"window_attn_kernel<10> ws=12 shift={shift} (Swin-L stage 1, B=4)", "ms": round(ms, 3),
                    "algorithmic_GBs": round(nb / ms / 1e6, 1), "frac_of_hbm_peak": round(nb / ms / 1e6 / 8000, 3)})


if __name__ == "__main__":
    main()
