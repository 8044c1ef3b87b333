"""Random-shape sweep of the kernel SOURCES (through tests/simt, on CPU) against the plain torch restatement of each operator
(tests/ops_emulation.py): shapes the fixed parity checks do not visit -- single rows, lengths around every tile boundary, strides,
masks that empty whole tiles, key splits with empty splits -- drawn from a seeded generator (reproducible; MQ_SIMT_FULL=1: 5x the
draws).  Shipped kernels and the opt-in ones (MQ_ATTN_RESIDENT, MQ_LN_VARIANT, MQ_OFFSET_CONV_VARIANT) go through the same draws.
TEST INFRASTRUCTURE ONLY (see tests/test_simt_kernels_cpu.py)."""
import os
import random
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

_CXX = os.environ.get("SIMT_CXX", "/opt/rocm/lib/llvm/bin/clang++")
pytestmark = pytest.mark.skipif(not os.path.exists(_CXX), reason=f"{_CXX} not found: the kernel-source emulation cannot be built here")
N_DRAWS = int(os.environ.get("MQ_SIMT_DRAWS", "5" if os.environ.get("MQ_SIMT_FULL", "0") == "1" else "1"))     # multiplier of the draw counts
TOL = 2e-3
SEED = int(os.environ.get("MQ_SIMT_SEED", "0"))                                # offset of every generator seed: other draws


@pytest.fixture(scope="module")
def ops():
    import simt
    with simt.installed() as o:
        yield o


def _close(got, ref, what, tol=TOL):
    got, ref = got.float(), ref.float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = float((got - ref).abs().max()) if ref.numel() else 0.0
    scale = max(1.0, float(ref.abs().max()) if ref.numel() else 1.0)
    assert err == err and err <= tol * scale, f"{what}: max |err| {err:.3e} vs scale {scale:.3e}"


def _edge(rng, tiles, hi):
    """a length at or next to a multiple of one of `tiles`, or anything in [1, hi]"""
    if rng.random() < 0.6:
        t = rng.choice(tiles)
        return max(1, min(hi, t * rng.randint(1, max(1, hi // t)) + rng.choice((-1, 0, 1))))
    return rng.randint(1, hi)


@pytest.mark.parametrize("variant", ["streaming", "resident+chunked"])
def test_attention_random_shapes(ops, monkeypatch, variant):
    import ops_emulation as emu
    monkeypatch.setenv("MQ_ATTN_RESIDENT", "1" if variant != "streaming" else "0")
    rng = random.Random(101 + SEED)
    g = torch.Generator().manual_seed(101 + SEED)
    for it in range(14 * N_DRAWS):
        B, H, D = rng.randint(1, 3), rng.randint(1, 4), rng.choice((32, 64))
        Nq, Nk = _edge(rng, (16, 32, 128), 300), _edge(rng, (8, 16, 64, 256), 700)
        nsplit = rng.choice((1, 1, 2, 3, 5)) if Nk > 64 else 1
        clamp = rng.choice((0.0, 0.0, 50000.0))
        q = (torch.randn(B, Nq, H * D, generator=g) * rng.choice((1.0, 3.0))).half()
        k, v = torch.randn(B, Nk, H * D, generator=g).half(), torch.randn(B, Nk, H * D, generator=g).half()
        kb = kl = None
        if rng.random() < 0.6:                                       # padding-style mask: a tail of masked keys per batch item
            kb, kl = torch.zeros(B, Nk), torch.zeros(B, dtype=torch.int32)
            for b in range(B):
                n = rng.randint(1, Nk)
                kb[b, n:] = -1e30
                kl[b] = n
            if rng.random() < 0.5 and Nk > 1:
                kb[:, rng.randrange(1, Nk)] = -1e30                  # one more masked key, possibly inside the valid range (key 0 stays)
                kl = None if rng.random() < 0.5 else kl              # (kv_len only promises that keys >= kv_len are masked by the bias)
            if kl is not None and rng.random() < 0.3:
                kl = None
        vt = F.pad(v, (0, 0, 0, (-Nk) % 8)).transpose(1, 2).contiguous()
        ref = emu.attention4(q.view(B, Nq, H, D), k.view(B, Nk, H, D), vt.view(B, H, D, -1), kb, None, clamp, nk=Nk)
        got = ops.attention(q, k, vt, H, D, key_bias=kb, clamp=clamp, nsplit=nsplit, nk=Nk, kv_len=kl)
        _close(got, ref, f"attention[{variant}] draw {it}: B={B} H={H} D={D} Nq={Nq} Nk={Nk} nsplit={nsplit} clamp={clamp} mask={kb is not None} kvlen={kl is not None}")


@pytest.mark.parametrize("variant", ["1", "2"])
def test_layernorm_random_shapes(ops, monkeypatch, variant):
    import ops_emulation as emu
    monkeypatch.setenv("MQ_LN_VARIANT", variant)
    rng = random.Random(202 + SEED)
    g = torch.Generator().manual_seed(202 + SEED)
    for it in range(16 * N_DRAWS):
        C = 8 * rng.choice((1, 2, 12, 16, 17, 24, 32, 33, 48, 64, 65, 96, 128, 129, 192, 256, 257, 384))
        rows = _edge(rng, (4, 8, 16, 64), 300)
        x = torch.randn(rows, C, generator=g) * 2 + 0.3
        x = x if rng.random() < 0.5 else x.half()
        res = None
        if rng.random() < 0.6:
            res = torch.randn(rows, C, generator=g)
            res = res if rng.random() < 0.5 else res.half()
        w, b = (torch.randn(C, generator=g) * 0.1 + 1).half(), (torch.randn(C, generator=g) * 0.1).half()
        kw = dict(residual=res, want_sum=rng.random() < 0.7, want_y32=rng.random() < 0.5)
        ref, got = emu.layer_norm(x, w, b, 1e-5, **kw), ops.layer_norm(x, w, b, 1e-5, **kw)
        ref, got = (ref if isinstance(ref, tuple) else (ref,)), (got if isinstance(got, tuple) else (got,))
        assert len(ref) == len(got)
        for i, (r, o) in enumerate(zip(ref, got)):
            assert r.dtype == o.dtype
            _close(o, r, f"layer_norm[v{variant}] draw {it} out {i}: rows={rows} C={C} x={x.dtype} res={None if res is None else res.dtype}")


def test_vlfuse_random_shapes(ops):
    import ops_emulation as emu
    rng = random.Random(303 + SEED)
    g = torch.Generator().manual_seed(303 + SEED)
    for it in range(6 * N_DRAWS):
        B, Hh = rng.randint(1, 3), rng.choice((4, 8))
        N, T = _edge(rng, (16, 64, 128), 400), 8 * rng.randint(1, 32)
        kv = None if rng.random() < 0.4 else torch.tensor([rng.randint(1, T) for _ in range(B)], dtype=torch.int32)
        v_ln = torch.randn(B, N, 256, generator=g).half()
        kf = (torch.randn(B, Hh, T, 256, generator=g) / 8).half()
        vo = torch.randn(B, Hh, T, 256, generator=g).half()
        if it % 2:                                       # the pipeline's operands: views of ONE projection output [B, T, heads*256 | heads*256 | 16]
            pr = torch.zeros(B, T, 2 * Hh * 256 + 16, dtype=torch.float16)
            pr[..., :Hh * 256] = kf.permute(0, 2, 1, 3).reshape(B, T, -1)
            pr[..., Hh * 256:2 * Hh * 256] = vo.permute(0, 2, 1, 3).reshape(B, T, -1)
            kf = pr[..., :Hh * 256].unflatten(-1, (Hh, 256)).permute(0, 2, 1, 3)
            vo = pr[..., Hh * 256:2 * Hh * 256].unflatten(-1, (Hh, 256)).permute(0, 2, 1, 3)
            assert not kf.is_contiguous()
        bias = torch.randn(B, Hh, T, generator=g)
        if kv is not None:
            for b in range(B):
                bias[b, :, int(kv[b]):] = -1e30                      # the caller's bias masks the keys beyond kv_len
        if T > 2:
            bias[:, :, rng.randrange(T // 2)] = -1e30               # and one key inside the valid range
            bias[:, :, T // 2 if (kv is None or int(kv.min()) > T // 2) else 0] = 0.0
            if kv is not None:
                for b in range(B):
                    if bool((bias[b, :, :int(kv[b])] < -1e29).all()):
                        bias[b, :, 0] = 0.0
        ob = torch.randn(256, generator=g).half()
        ref = emu.vlfuse_i2t(v_ln.float(), kf.float(), vo.float(), bias, ob.float(), kv, 0)
        got = ops.vlfuse_i2t(v_ln, kf, vo, bias, ob, kv, max_kv=0 if kv is None else int(kv.max()))
        _close(got, ref, f"vlfuse_i2t draw {it}: B={B} heads={Hh} N={N} T={T} kv={None if kv is None else kv.tolist()}")
        ns = rng.randint(1, 4)
        ref = emu.vlfuse_t2i(kf.float(), v_ln.float(), ns, kv_len=kv)
        got = ops.vlfuse_t2i(kf, v_ln, ns, kv_len=kv)
        live = torch.ones(B, T, dtype=torch.bool)
        if kv is not None:                                            # rows of all-padding 128-row tiles come back as zeros by contract
            for b in range(B):
                live[b, -(-int(kv[b]) // 128) * 128:] = False
        _close(got[live], ref[live], f"vlfuse_t2i draw {it}: B={B} heads={Hh} N={N} T={T} nsplit={ns}")


@pytest.mark.parametrize("variant", ["1", "2"])
def test_conv_and_dcn_random_shapes(ops, monkeypatch, variant):
    import ops_emulation as emu
    monkeypatch.setenv("MQ_OFFSET_CONV_VARIANT", variant)
    rng = random.Random(404 + SEED)
    g = torch.Generator().manual_seed(404 + SEED)
    for it in range(5 * N_DRAWS):
        B, H, W = rng.randint(1, 2), _edge(rng, (8,), 27), _edge(rng, (16,), 37)
        C = rng.choice((64, 128, 256))
        x = torch.randn(B, H, W, C, generator=g).half()
        w27 = torch.zeros(32, 9 * C, dtype=torch.float16)
        w27[:27] = (torch.randn(27, 9 * C, generator=g) / 48).half()
        b27 = torch.randn(27, generator=g).half()
        _close(ops.conv3x3_nchw32(x, w27, b27, 27), emu.conv3x3_nchw32(x, w27, b27, 27), f"offset conv[v{variant}] draw {it}: {B}x{H}x{W}x{C}")
        if variant == "2" or C != 256:
            continue
        stride = rng.choice((1, 2))
        w = (torch.randn(256, 9 * C, generator=g) / 48).half()
        bias = torch.randn(256, generator=g).half()
        _close(ops.conv3x3(x, w, bias, 256, stride), emu.conv3x3(x, w, bias, 256, stride), f"conv3x3 draw {it}: {B}x{H}x{W} s{stride}", 3e-3)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        om = torch.randn(B, 27, Ho, Wo, generator=g) * rng.choice((0.3, 2.0, 40.0))      # up to offsets far outside the image
        y, hw = ops.dcnv2(x, om.contiguous(), w, bias, stride)
        yr, hwr = emu.dcnv2(x, om, w, bias, stride)
        assert tuple(hw) == tuple(hwr)
        _close(y, yr, f"dcnv2 draw {it}: {B}x{H}x{W} s{stride}", 4e-3)


def test_scoring_and_nms_random_shapes(ops, monkeypatch):
    import ops_emulation as emu
    rng = random.Random(505 + SEED)
    g = torch.Generator().manual_seed(505 + SEED)
    for it in range(8 * N_DRAWS):
        B, HW, T = rng.randint(1, 3), _edge(rng, (4, 64), 200), rng.choice((16, 100, 255, 256))
        L, MT = _edge(rng, (64,), 90), rng.randint(1, 5)
        dot = (torch.randn(B, HW, T, generator=g) * 2)
        dot = dot if rng.random() < 0.5 else dot.half()
        tb, ctr = torch.randn(B, T, generator=g), torch.randn(B, HW, generator=g).half()
        tok = torch.full((L, MT), -1, dtype=torch.int32)
        for l in range(L):
            n = rng.randint(0, MT)
            tok[l, :n] = torch.tensor(rng.sample(range(T), n), dtype=torch.int32)
        for agg in (0, 1, 2):
            r, c = ops.align_scores(dot, tb, tok, ctr, 0.05, want_cls=True, agg=agg)
            rr, cr = emu.align_scores(dot, tb, tok, ctr, 0.05, want_cls=True, agg=agg)
            _close(c, cr, f"align_scores cls draw {it} agg={agg}: B={B} HW={HW} T={T} L={L} MT={MT}", 1e-5)
            far = (cr - 0.05).abs() > 1e-5
            _close(r[far], rr[far], f"align_scores ranked draw {it} agg={agg}", 1e-5)
    for it in range(4 * N_DRAWS):
        B, N = rng.randint(1, 3), _edge(rng, (64,), 900)
        xy = torch.rand(B, N, 2, generator=g) * 200
        boxes = torch.cat([xy, xy + 10 + torch.rand(B, N, 2, generator=g) * 60], -1).contiguous()
        labels = torch.randint(1, 4, (B, N), generator=g, dtype=torch.int32)
        nvalid = torch.tensor([rng.randint(0, N) for _ in range(B)], dtype=torch.int32)
        monkeypatch.setenv("MQ_NMS_EARLY_STOP", "0")
        keep = ops.ml_nms(boxes, labels, nvalid, 0.6)
        ref = emu.ml_nms(boxes, labels, nvalid, 0.6)
        assert torch.equal(keep, ref), f"ml_nms draw {it}: B={B} N={N} nvalid={nvalid.tolist()}"
        monkeypatch.setenv("MQ_NMS_EARLY_STOP", "1")
        K = rng.randint(1, 200)
        part = ops.ml_nms(boxes, labels, nvalid, 0.6, max_keep=K)
        for b in range(B):
            kf, kp = keep[b].nonzero().flatten(), part[b].nonzero().flatten()
            n = min(K, len(kf))
            assert torch.equal(kp[:n], kf[:n]) and bool((part[b] <= keep[b]).all()), f"ml_nms_topk draw {it} K={K}"


def test_sparse_attention_and_window_attention_random_shapes(ops):
    import ops_emulation as emu
    rng = random.Random(606 + SEED)
    g = torch.Generator().manual_seed(606 + SEED)
    for it in range(5 * N_DRAWS):
        B, T, V, S = rng.randint(1, 2), _edge(rng, (32,), 80), rng.randint(1, 60), rng.choice((1, 5, 8, 9, 16, 17, 25))
        q, kv = torch.randn(B, T, 512, generator=g).half(), torch.randn(B, V, 1024, generator=g).half()
        idx = torch.full((B, T, S), -1, dtype=torch.int32)
        for b in range(B):
            for t in range(T):
                n = rng.choice((0, 0, S, rng.randint(0, S)))
                n = min(n, V)
                idx[b, t, :n] = torch.tensor(rng.sample(range(V), n), dtype=torch.int32)
        _close(ops.gcp_sparse_attention(q, kv, idx), emu.gcp_sparse_attention(q, kv, idx), f"gcp_sparse draw {it}: B={B} T={T} V={V} S={S}")
    for it in range(4 * N_DRAWS):
        ws, heads = rng.choice(((7, 3), (7, 6), (12, 3)))
        C = heads * 32
        B, H, W = rng.randint(1, 2), rng.randint(1, 2 * ws + 3), rng.randint(1, 2 * ws + 3)
        shift = rng.choice((0, ws // 2))
        qkv = torch.randn(B, H, W, 3 * C, generator=g).half()
        qb = torch.randn(3 * C, generator=g).half()
        rel = torch.randn(heads, ws * ws, ws * ws, generator=g)
        _close(ops.window_attention(qkv, qb, rel, heads, ws, shift), emu.window_attention(qkv, qb, rel, heads, ws, shift),
               f"window_attention draw {it}: B={B} {H}x{W} ws={ws} heads={heads} shift={shift}", 3e-3)


def test_round3_fused_operators_random_shapes(ops):
    """mq_window_attn_qkv_fwd (both widths: resident and streamed weights; images smaller than a window, several trips of the persistent
    workgroups, idle waves), mq_dyrelu_ln_fwd (1 .. 6 levels of ragged sizes) and mq_swin_mlp2_fwd across the pass / tail split."""
    import math
    import ops_emulation as emu
    rng = random.Random(909 + SEED)
    g = torch.Generator().manual_seed(909 + SEED)
    for it in range(6 * N_DRAWS):
        heads = rng.choice((3, 6))
        C, ws = heads * 32, 7
        B, H, W = rng.randint(1, 3), rng.randint(1, 5 * ws + 3), rng.randint(1, 5 * ws + 3)
        shift = rng.choice((0, ws // 2))
        x = torch.randn(B, H, W, C, generator=g).half()
        w = (torch.randn(3 * C, C, generator=g) / math.sqrt(C)).half()
        bias = (torch.randn(3 * C, generator=g) * 0.2).half()
        rel = torch.randn(heads, ws * ws, ws * ws, generator=g) * 0.3
        _close(ops.window_attention_qkv(x, w, bias, rel, heads, ws, shift), emu.window_attention_qkv(x, w, bias, rel, heads, ws, shift),
               f"window_attention_qkv draw {it}: B={B} {H}x{W} C={C} shift={shift}", 4e-3)
    for it in range(6 * N_DRAWS):
        nl = rng.randint(1, 6)
        sizes = [(rng.randint(1, 9), rng.randint(1, 11)) for _ in range(nl)]
        B, N = rng.randint(1, 3), sum(h * w_ for h, w_ in sizes)
        big = torch.randn(B, N + 5, 256, generator=g).half() * 2          # rows of a larger buffer: batch stride != N * C
        x = big[:, 2:2 + N]
        coef = torch.randn(nl, B, 4, 256, generator=g)
        gam, bet = (torch.randn(256, generator=g) * 0.1 + 1).half(), (torch.randn(256, generator=g) * 0.1).half()
        _close(ops.dyrelu_layer_norm(x, coef, sizes, gam, bet, 1e-5), emu.dyrelu_layer_norm(x, coef, sizes, gam, bet, 1e-5),
               f"dyrelu_layer_norm draw {it}: B={B} sizes={sizes}", 4e-3)
    for it in range(4 * N_DRAWS):
        C = rng.choice((96, 192, 384))
        # the emulator's "chip" has 4 CUs: 16 / 12 / 4 workgroup slots of 64 / 64 / 128 tokens -> lengths around one and two passes
        slot = {96: 16 * 64, 192: 12 * 64, 384: 4 * 128}[C]
        M = max(1, slot * rng.choice((1, 1, 2)) + rng.choice((-17, -1, 0, 1, 15, 16, 33, 70)))
        x = torch.randn(M, C, generator=g)
        delta = (torch.randn(M, C, generator=g) * 0.5).half() if rng.random() < 0.7 else None
        lg_, lb_ = (torch.randn(C, generator=g) * 0.1 + 1).half(), (torch.randn(C, generator=g) * 0.1).half()
        w1, b1 = (torch.randn(4 * C, C, generator=g) / math.sqrt(C)).half(), (torch.randn(4 * C, generator=g) * 0.1).half()
        w2, b2 = (torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C)).half(), (torch.randn(C, generator=g) * 0.1).half()
        w1f, w2f = ops.swin_mlp2_pack(w1, w2)
        nln = (lg_, lb_, 1e-5) if rng.random() < 0.7 else None
        flags = rng.choice((0, 2, 1, 4))
        got = ops.swin_mlp2(x, delta, lg_, lb_, 1e-5, w1f, b1, w2f, b2, next_ln=nln, flags=flags)
        ref = emu.swin_mlp2(x, delta, lg_, lb_, 1e-5, w1f, b1, w2f, b2, next_ln=nln)
        if nln is None:
            got, ref = (got,), (ref,)
        for a_, b_, what in zip(got, ref, ("out", "next LN")):
            _close(a_, b_, f"swin_mlp2 draw {it}: C={C} M={M} flags={flags} delta={delta is not None}: {what}", 3e-3)


def test_round4_operators_random_shapes(ops):
    """mq_attn_text_fwd (caption lengths around every 16-key block and the 160-key variant switch, per-item kv_len, max_kv above / at / absent,
    clamp, D = 32 / 64), mq_patch_embed_fwd (both pixel layouts, widths around multiples of 16 patches, C = 96 / 192), and the
    post-processing kernels (mq_post_select_fwd over one / several slices with scores quantised so that ties straddle every cut,
    mq_post_sort_fwd, mq_post_finalize_fwd) against their torch restatements."""
    import ops_emulation as emu
    rng = random.Random(4004 + SEED)
    g = torch.Generator().manual_seed(4004 + SEED)
    for it in range(8 * N_DRAWS):
        B, H, D = rng.randint(1, 3), rng.randint(1, 4), rng.choice((32, 64))
        T = rng.choice((256, 256, 8 * rng.randint(1, 32)))
        kv = max(1, min(T, _edge(rng, (16, 32, 160), T)))
        clamp = rng.choice((0.0, 0.0, 50000.0, 2.5))
        qkv = (torch.randn(B, T, 3 * H * D, generator=g) * rng.choice((1.0, 4.0))).half()
        kl = torch.tensor([max(1, kv - rng.randint(0, 20) * (b > 0)) for b in range(B)], dtype=torch.int32)
        kb = torch.zeros(B, T)
        for b in range(B):
            kb[b, int(kl[b]):] = -1e30
        mk = rng.choice((0, kv, min(T, kv + rng.randint(0, 40))))
        got = ops.attention_text(qkv, H, key_bias=kb, clamp=clamp, kv_len=kl if rng.random() < 0.8 else None, max_kv=mk)
        _close(got, emu.attention_text(qkv, H, key_bias=kb, clamp=clamp), f"attention_text draw {it}: B={B} H={H} D={D} T={T} kv={kv} max_kv={mk} clamp={clamp}")
    for it in range(6 * N_DRAWS):
        C = rng.choice((96, 192))
        B, Hi, Wi = rng.randint(1, 2), 4 * rng.randint(1, 9), 4 * _edge(rng, (16,), 70)
        img = torch.randn(B, 3, Hi, Wi, generator=g).half()
        w = (torch.randn(C, 3, 4, 4, generator=g) * 0.2).half()
        prm = [torch.randn(C, generator=g) * s_ + o_ for s_, o_ in ((0.1, 0), (0.2, 1), (0.1, 0), (0.2, 1), (0.1, 0))]
        if rng.random() < 0.5:
            pix, wpk = img.float().contiguous(), ops.patch_embed_pack(w.float(), nchw=True).half()
        else:
            pix, wpk = img.permute(0, 2, 3, 1).contiguous(), ops.patch_embed_pack(w.float()).half()
        for a_, b_, what in zip(ops.patch_embed(pix, wpk, *prm), emu.patch_embed(pix, wpk, *prm), ("stream", "norm1")):
            _close(a_, b_, f"patch_embed draw {it}: B={B} {Hi}x{Wi} C={C} {'fp32 NCHW' if pix.dtype == torch.float32 else 'NHWC'}: {what}", 2e-3)
    for it in range(5 * N_DRAWS):
        B, L, nl = rng.randint(1, 2), rng.randint(1, 12), rng.randint(1, 4)
        shapes = [(rng.randint(1, 40), rng.randint(1, 90)) for _ in range(nl)]
        if it % 2 == 0:
            shapes[0] = (rng.randint(60, 75), rng.randint(60, 75))           # > 32768 scores with L >= 8: several slices
            L = max(L, 8)
        topn = rng.choice((1, 17, 300, 1000, 1500))
        quant = rng.choice((None, 3, 40))
        dens = rng.choice((0.0, 0.02, 0.5, 1.0))
        ranked, reg, anchors, ks = [], [], [], []
        for (h, w_) in shapes:
            hw = h * w_
            v = torch.rand(B, hw, L, generator=g)
            if quant:
                v = (torch.floor(v * quant) + 1) / (quant + 1)
            ranked.append(torch.where(torch.rand(B, hw, L, generator=g) < dens, v, torch.full_like(v, -1.0)).contiguous())
            reg.append((torch.randn(B, hw, 4, generator=g) * 2).contiguous())
            xy = torch.rand(hw, 2, generator=g) * 300
            anchors.append(torch.cat([xy, xy + 8 + torch.rand(hw, 2, generator=g) * 64], 1).contiguous())
            ks.append(min(topn, hw * L))
        lab = torch.randperm(L, generator=g).to(torch.int32) + 1
        wh = torch.tensor([[333.0, 250.0]] * B)
        assert ops.post_select_supported([h * w_ for h, w_ in shapes], ks, B, L)
        gb, gs, gl, gi = ops.post_select(ranked, reg, anchors, ks, lab, wh)
        eb, es, el, ei = emu.post_select(ranked, reg, anchors, ks, lab.long(), wh)
        what = f"post draw {it}: B={B} shapes={shapes} L={L} k={ks} quant={quant} dens={dens}"
        assert torch.equal(gi, ei) and torch.equal(gl.int(), el.int()), what + ": candidate ids / labels"
        _close(gs, es, what + ": scores", 2e-7)
        _close(gb, eb, what + ": boxes", 1e-6)
        hb, hs, hl, hn = ops.post_sort(gb, gs, gl, ks)
        sb, ss, sl, sn = emu.post_sort(gb, gs, gl, ks)
        assert torch.equal(hs, ss) and torch.equal(hl.int(), sl.int()) and torch.equal(hn.int(), sn.int()) and torch.equal(hb, sb), what + ": merge"
        tot = sum(ks)
        K = rng.randint(1, tot)
        K2 = min(tot, K + rng.choice((0, 1, 16)))
        keep = (torch.rand(B, tot, generator=g) < rng.choice((0.1, 0.7, 1.0))).to(torch.uint8)
        ho, hc = ops.post_finalize(hb, hs, hl, keep, K, K2)
        fo, fc = emu.post_finalize(hb, hs, hl, keep, K, K2)
        assert torch.equal(hc.int(), fc.int()) and torch.equal(ho, fo), what + f": finalize K={K} K2={K2}"


def test_grouped_dyconv_kernels_random_pyramids(ops):
    """mq_conv3x3_nchw32_group_fwd and mq_dyconv_epilogue_group on random pyramids (1 .. 6 levels, level sizes around the 8 x 16 tile and
    the 128-position block edges, levels as slices of one token buffer, B = 1 .. 3): the conv against the per-level kernel (fp32 summation
    order apart) and F.conv2d; the epilogue against the per-level launches (bit for bit) with every branch mix (1 .. 3 direct branches,
    with / without a coarser bilinear one)."""
    import torch.nn.functional as F
    rng = random.Random(5005 + SEED)
    g = torch.Generator().manual_seed(5005 + SEED)
    for it in range(3 * N_DRAWS):
        B, nl = rng.randint(1, 3), rng.randint(1, 6)
        sizes = [(_edge(rng, (8, 16), 24), _edge(rng, (16, 32), 40)) for _ in range(nl)]
        tok = torch.randn(B, sum(h * w_ for h, w_ in sizes) + 3, 256, generator=g).half()
        w = (torch.randn(27, 256, 3, 3, generator=g) / 48).half()
        bias = torch.randn(27, generator=g).half()
        wp = torch.cat([w.permute(0, 2, 3, 1).reshape(27, -1), torch.zeros(5, 9 * 256, dtype=torch.float16)], 0).contiguous()
        lv, off = [], 3
        for (h, w_) in sizes:
            lv.append(tok[:, off:off + h * w_].reshape(B, h, w_, 256))
            off += h * w_
        got = ops.conv3x3_nchw32_group(lv, wp, bias, 27)
        for l, (x, y) in enumerate(zip(lv, got)):
            what = f"offset conv group draw {it}: B={B} sizes={sizes} level {l}"
            _close(y, ops.conv3x3_nchw32(x, wp, bias, 27), what + " vs the per-level kernel", 2e-6)
            _close(y, F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias.float(), padding=1), what + " vs F.conv2d", 1e-5)
    for it in range(3 * N_DRAWS):
        B, nl = rng.randint(1, 3), rng.randint(1, 5)
        sizes = [(rng.randint(1, 14), _edge(rng, (16,), 20)) for _ in range(nl)]
        out_g = torch.zeros(B, sum(h * w_ for h, w_ in sizes), 256, dtype=torch.float16)
        out_p = torch.zeros_like(out_g)
        w0, b0 = (torch.randn(64, 256, generator=g) / 16).half(), (torch.randn(64, generator=g) * 0.1).half()
        w2, b2 = (torch.randn(1024, 64, generator=g) / 8).half(), (torch.randn(1024, generator=g) * 0.1).half()
        levels, off = [], 0
        for (h, w_) in sizes:
            branches = []
            for _ in range(rng.randint(0, 3) if rng.random() < 0.7 else 0):
                branches.append(((torch.randn(B, h * w_, 256, generator=g)).half(), torch.randn(B, 256, 2, generator=g) * 0.5, h, w_))
            if not branches or (len(branches) < 3 and rng.random() < 0.6):
                hs, ws = max(1, (h + 1) // 2), max(1, (w_ + 1) // 2)
                if (hs, ws) == (h, w_) or len(branches) == 0 and rng.random() < 0.3:
                    branches.append((torch.randn(B, h * w_, 256, generator=g).half(), torch.randn(B, 256, 2, generator=g) * 0.5, h, w_))
                else:
                    branches.insert(rng.randint(0, len(branches)), (torch.randn(B, hs * ws, 256, generator=g).half(), torch.randn(B, 256, 2, generator=g) * 0.5, hs, ws))
            levels.append((branches, h, w_, off))
            off += h * w_
        rc = torch.zeros(nl, B, 4, 256)
        ops.dyconv_epilogue_group([(br, h, w_, out_g[:, o:o + h * w_]) for br, h, w_, o in levels], w0, b0, w2, b2, rc)
        for l, (br, h, w_, o) in enumerate(levels):
            _, pool = ops.dyconv_fuse(br, h, w_, out=out_p[:, o:o + h * w_])
            ref = ops.dyrelu_coef(pool, h * w_, w0, b0, w2, b2)
            assert torch.equal(rc[l], ref), f"epilogue group draw {it}: sizes={sizes} level {l}: DYReLU coefficients"
        assert torch.equal(out_g, out_p), f"epilogue group draw {it}: sizes={sizes} branches={[len(lv_[0]) for lv_ in levels]}"


def test_swin_mlp_roi_align_and_msdeform_random_shapes(ops):
    import math
    import ops_emulation as emu
    rng = random.Random(707 + SEED)
    g = torch.Generator().manual_seed(707 + SEED)
    for it in range(4 * N_DRAWS):
        C, M = rng.choice((96, 192, 384)), _edge(rng, (16, 32, 64, 128), 400)
        x = torch.randn(M, C, generator=g) * 1.5
        delta = (torch.randn(M, C, generator=g) * 0.5).half() if rng.random() < 0.7 else None
        lg, lb = (torch.randn(C, generator=g) * 0.1 + 1).half(), (torch.randn(C, generator=g) * 0.1).half()
        w1, b1 = (torch.randn(4 * C, C, generator=g) / math.sqrt(C)).half(), (torch.randn(4 * C, generator=g) * 0.1).half()
        w2, b2 = (torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C)).half(), (torch.randn(C, generator=g) * 0.1).half()
        w1f, w2f = ops.swin_mlp2_pack(w1, w2)
        nxt = ((torch.randn(C, generator=g) * 0.1 + 1).half(), (torch.randn(C, generator=g) * 0.1).half(), 1e-5) if rng.random() < 0.6 else None
        flags = rng.choice((0, 1, 2, 4))
        got, ref = ops.swin_mlp2(x, delta, lg, lb, 1e-5, w1f, b1, w2f, b2, next_ln=nxt, flags=flags), emu.swin_mlp2(x, delta, lg, lb, 1e-5, w1f, b1, w2f, b2, next_ln=nxt)
        got, ref = (got if isinstance(got, tuple) else (got,)), (ref if isinstance(ref, tuple) else (ref,))
        for i, (a, b) in enumerate(zip(got, ref)):
            _close(a, b, f"swin_mlp2 draw {it} out {i}: C={C} M={M} flags={flags} delta={delta is not None} next={nxt is not None}", 2e-3 if i else 1e-3)
    for it in range(4 * N_DRAWS):
        N, C, H, W = rng.randint(1, 2), rng.choice((8, 64, 256)), rng.randint(1, 30), rng.randint(1, 40)
        feat = torch.randn(N, C, H, W, generator=g)
        f16 = feat.half().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)                 # NHWC memory, NCHW view
        R = rng.randint(1, 9)
        x1, y1 = torch.rand(R, generator=g) * W * 16 - 24, torch.rand(R, generator=g) * H * 16 - 24     # partly outside the image
        rois = torch.stack([torch.randint(0, N, (R,), generator=g).float(), x1, y1, x1 + torch.rand(R, generator=g) * 200,
                            y1 + torch.rand(R, generator=g) * 200], 1)
        for aligned in (False, True):
            sr = rng.choice((0, 2))
            for f in (f16, feat):
                _close(ops.roi_align(f, rois, 7, 1.0 / 16, sr, aligned=aligned), emu.roi_align(f, rois, 7, 1.0 / 16, sr, aligned=aligned),
                       f"roi_align draw {it}: {N}x{C}x{H}x{W} R={R} aligned={aligned} sr={sr} {f.dtype}", 1e-4)
    for it in range(3 * N_DRAWS):
        B, M, D = rng.randint(1, 2), 8, 32
        shapes = [(rng.randint(2, 14), rng.randint(2, 18)) for _ in range(4)]
        S, Q = sum(h * w for h, w in shapes), _edge(rng, (4, 64), 150)
        value = torch.randn(B, S, M * D, generator=g).half()
        qp = torch.cat([torch.randn(B, Q, M * 16 * 2, generator=g) * 3.0, torch.randn(B, Q, M * 16, generator=g)], -1).half()
        nd = rng.choice((2, 4))
        ref_pts = torch.rand(B, Q, 4, nd, generator=g) * 1.2 - 0.1                                   # some reference points outside [0, 1]
        if nd == 4:
            ref_pts[..., 2:] = ref_pts[..., 2:].abs() * 0.3 + 0.02
        vhw = None
        if rng.random() < 0.5:
            vhw = torch.tensor([[[rng.randint(1, h), rng.randint(1, w)] for (h, w) in shapes] for _ in range(B)], dtype=torch.int32)
        _close(ops.ms_deform_attn_q(value, shapes, qp, ref_pts, M, valid_hw=vhw), emu.ms_deform_attn_q(value, shapes, qp, ref_pts, M, valid_hw=vhw),
               f"ms_deform_attn_q draw {it}: B={B} shapes={shapes} Q={Q} ref_dim={nd} valid_hw={vhw is not None}")


def test_bf16_twins_random_shapes(ops, monkeypatch):
    """the *_bf16 entry points on a few of the same draws (tolerance x 8)"""
    import ops_emulation as emu
    rng = random.Random(808 + SEED)
    g = torch.Generator().manual_seed(808 + SEED)
    bf = torch.bfloat16
    for res_attn in ("0", "1"):
        monkeypatch.setenv("MQ_ATTN_RESIDENT", res_attn)
        for it in range(5 * N_DRAWS):
            B, H, D = rng.randint(1, 2), rng.randint(1, 3), rng.choice((32, 64))
            Nq, Nk = _edge(rng, (16, 32, 128), 200), _edge(rng, (8, 16, 64, 256), 500)
            q, k, v = (torch.randn(B, n, H * D, generator=g).to(bf) for n in (Nq, Nk, Nk))
            vt = F.pad(v, (0, 0, 0, (-Nk) % 8)).transpose(1, 2).contiguous()
            ref = emu.attention4(q.view(B, Nq, H, D), k.view(B, Nk, H, D), vt.view(B, H, D, -1), None, None, 0.0, nk=Nk)
            got = ops.attention(q, k, vt, H, D, nk=Nk)
            assert got.dtype == bf
            _close(got, ref, f"attention bf16 (resident={res_attn}) draw {it}: B={B} H={H} D={D} Nq={Nq} Nk={Nk}", 8 * TOL)
    for variant in ("1", "2"):
        monkeypatch.setenv("MQ_LN_VARIANT", variant)
        for it in range(5 * N_DRAWS):
            C, rows = 8 * rng.choice((12, 24, 32, 48, 96, 192)), _edge(rng, (4, 16), 200)
            x, res = torch.randn(rows, C, generator=g) * 2, torch.randn(rows, C, generator=g).to(bf)
            w, b = (torch.randn(C, generator=g) * 0.1 + 1).to(bf), (torch.randn(C, generator=g) * 0.1).to(bf)
            ref, got = emu.layer_norm(x, w, b, 1e-5, residual=res, want_y32=True), ops.layer_norm(x, w, b, 1e-5, residual=res, want_y32=True)
            for i, (r, o) in enumerate(zip(ref, got)):
                assert r.dtype == o.dtype
                _close(o, r, f"layer_norm bf16 [v{variant}] draw {it} out {i}: rows={rows} C={C}", 8 * TOL)
