"""HIP path vs CPU oracle parity checks (run on the GPU box; used by tests/test_gpu_parity.py, by
tests/gpu_diag.py which prints the whole table, and by __graft_entry__.smoke()).

Tolerance (stated here once): the kernels take fp16 inputs and accumulate in fp32; the oracle is fp32 on the
same (fp16-rounded) inputs.  A check passes when   max|hip - ref| <= TOL * max(1, max|ref|)   with
TOL = 2e-3 (about two fp16 ulps at the output scale) for single kernels / blocks, and the looser, per-check
values given below for deep stacks (every layer re-rounds its activations to fp16; the reference itself runs
fp32).  Integer / index outputs (NMS keep sets, labels) must match exactly on well-separated inputs.
"""
import math

import torch
import torch.nn.functional as F

TOL = 2e-3
# tests/test_simt_kernels_cpu.py runs these checks on CPU tensors through the lane-by-lane emulation of the kernel sources:
# QUICK drops the largest cases of a few sweeps (minutes on the emulator), PINS = False skips the comparisons against the
# reference's own device kernels (oracle/_ref needs a GPU).  On the GPU box both keep their defaults.
QUICK = False
PINS = True
# The 16-bit operand type under test: fp16 (default; the entry points of include/mqdet_hip.h) or bf16 (their *_bf16 twins,
# MODEL.COMPUTE_DTYPE = "bfloat16", BASELINE.json configs[3]).  `use_dtype(torch.bfloat16)` switches every check below: inputs and
# weights are rounded to bf16, the product runs its bf16 kernels, and every tolerance is multiplied by 8 = 2^(11 - 8), the ratio of
# the two formats' rounding steps (fp16: 11 significant bits, bf16: 8).
H16 = torch.float16
TOL_SCALE = 1.0


ELEM_TOL = 1e-3      # atol = rtol of the element-wise column of _stat ("atol = rtol = 1e-3")
ELEM_FRAC_16 = 1e-5  # 16-bit operands, per-kernel rows: at most this fraction of elements outside atol + rtol |ref|
# ... except the kernels that round an INTERMEDIATE to 16 bits because it is the operand of their second MFMA -- the softmax probabilities P
# of every attention kernel (and q | k | v inside the Swin attention with the projection fused: the rounding point of the reference's qkv
# tensor).  P <= 1 carries a relative error of 2^-12; where one key holds the probability mass and the head outputs cancel in the sum over
# heads (VLFuse image side: 8 heads x |Vo| ~ 3 summed into an output of ~ 1) single elements land at 2 - 3e-3 absolute.  The fp32-operand
# build of the same kernels has ZERO violations in every row (the gate of the precise mode), so this is operand rounding, not logic.
ELEM_FRAC_P16 = 1e-3
_P16_ROWS = ("attn", "window_attn", "vlfuse", "gcp sparse", "gcp pre", "... vs GEMM + mq_window_attn_fwd")
ELEM_FRAC_DEEP_F32 = 1e-4   # fp32 operands, deep-stack rows: fraction of elements that may lie outside atol + rtol |ref| (per-kernel rows: none).
                            # Measured in the final GPU suite of round 5 (profiles/r05_final_gpu_suite_ladder.jsonl): ZERO in all 413 fp32-operand rows
F32_TOL = 1e-3       # the north-star tolerance: met by every stage once the operands are not rounded (tests/test_simt_fp32_operands_cpu.py)


def use_dtype(dtype):
    """torch.float32: the precise mode (MODEL.COMPUTE_DTYPE = "float32", the *_f32 entry points -- on the device, or through the
    emulation with `simt.installed(f32=...)`); the caller also selects KERNELS["F32_OPERANDS"] (environment MQ_F32_OPERANDS + ops.configure()).
    Nothing is rounded to 16 bits, EVERY tolerance becomes F32_TOL = 1e-3 of the reference's range and no element may lie outside
    atol = rtol = 1e-3."""
    global H16, TOL_SCALE
    assert dtype in (torch.float16, torch.bfloat16, torch.float32)
    H16, TOL_SCALE = dtype, (8.0 if dtype == torch.bfloat16 else 1.0)
    _CACHE.clear()


# VERDICT r3 item 8 -- every gate at max(1e-3, 2 x measured): tests/golden/device_measured.json holds, per row name, the normalised error
# the MI355X produced (tools/measured_gate.py from the ladder file of a whole GPU-suite run).  ON THE DEVICE a row found there is gated at
# min(stated tolerance, max(1e-3 x dtype scale, 2 x measured)): a kernel that gets 3 x worse than its last measurement fails whatever the
# stated per-family tolerance allows.  (The CPU emulation of the kernel sources rounds differently in exp / rcp: it keeps the stated values.)
# DEEP-STACK rows (whole models, whole fusion layers) get 4 x instead of 2 x: their max-norm error is a chaotic statistic -- between the round-3
# and the round-4 ladder (every per-kernel row equal or better) 75 of them moved DOWN by 30 % ... 10 x and a handful UP by up to 3.3 x
# ("[bf16] gdino[vq ...] BERT (+GCP) last hidden" 1.7e-2 -> 5.7e-2 with no kernel of its path changed: the Swin front feeding the
# pre-select rounds differently).  Per-kernel rows keep 2 x.
MEASURED_GATE = torch.cuda.is_available()
MEASURED_MARGIN, MEASURED_MARGIN_DEEP, MEASURED_FLOOR = 2.0, 4.0, 1e-3
_DEEP_ROWS = ("full:", "bench[", "gdino[", "fusion layer", "extract_query")
_MEASURED = None


def _measured(name):
    global _MEASURED
    if _MEASURED is None:
        import json
        import os
        f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "device_measured.json")
        _MEASURED = json.load(open(f))["norm_err"] if os.path.exists(f) else {}
    return _MEASURED.get(name)


def _stat(name, got, ref, tol=TOL, elem_gate=None):
    tol = tol * TOL_SCALE
    stated = tol                  # the check's own tolerance, before the measured gate below tightens it: decides whether the row is a per-kernel row
    if H16 == torch.bfloat16:
        name = "[bf16] " + name
    elif H16 == torch.float32:
        name, tol = "[fp32 operands] " + name, min(tol, F32_TOL)
    if MEASURED_GATE and tol > 0 and H16 != torch.float32:
        m = _measured(name)
        if m is not None:
            margin = MEASURED_MARGIN_DEEP if any(t in name for t in _DEEP_ROWS) else MEASURED_MARGIN
            tol = min(tol, max(MEASURED_FLOOR * TOL_SCALE, margin * m))
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    diff = (got - ref).abs()
    err = diff.max().item() if ref.numel() else 0.0
    scale = max(1.0, ref.abs().max().item() if ref.numel() else 1.0)
    mean = diff.mean().item() if ref.numel() else 0.0
    bad = not math.isfinite(err)
    # SURVEY 8(c) / VERDICT r4 item 1a -- the ELEMENT-WISE statement beside the max-norm one: the fraction of elements outside
    # |got - ref| <= atol + rtol |ref| with atol = rtol = 1e-3 (x 8 for bf16), i.e. what torch.allclose(got, ref, 1e-3, 1e-3) counts.
    e_tol = ELEM_TOL * TOL_SCALE
    viol = (diff > e_tol + e_tol * ref.abs()).float().mean().item() if ref.numel() else 0.0
    # gated: (a) with fp32 operands EVERY row, at zero violations; (b) with 16-bit operands the per-KERNEL rows (stated tolerance <= 2e-3 of
    # the range: one kernel against the oracle on the same rounded operands) at ELEM_FRAC_16 -- what a 16-bit-stored output may lose to
    # rounding of values that cancel; deep-stack rows (whole models / layers, 16-bit storage between kernels) carry the column ungated.
    deep = any(t_ in name for t_ in _DEEP_ROWS)
    if H16 == torch.float32:
        # per-kernel rows: none.  Deep rows (whole models / fusion layers): a stated bound of 1e-4 of the elements against box-to-box variance of
        # the fp32 summation order through 12 + 6 layers (the first device run of round 5, still with the hardware exp / rcp approximations in the
        # softmaxes, had 2.9e-3 of the P7 alignment logits outside -- max |err| 1.96e-3 on values up to 11.5; with library exp / IEEE division
        # the final suite has none in any row, worst stage 9.3e-5 of the range)
        e_ok = viol == 0.0 if not deep else viol <= ELEM_FRAC_DEEP_F32
    else:
        frac = ELEM_FRAC_P16 if any(name.startswith(t_) or name.startswith("[bf16] " + t_) for t_ in _P16_ROWS) else ELEM_FRAC_16
        gated = (not deep and stated <= 2e-3 * TOL_SCALE) if elem_gate is None else bool(elem_gate)
        e_ok = (not gated) or viol <= frac
    return {"name": name, "max_err": err, "mean_err": mean, "ref_absmax": scale, "norm_err": err / scale, "elem_viol_frac": viol,
            "elem_ok": e_ok, "tol": tol, "ok": (not bad) and err <= tol * scale and e_ok}


def _ref_attention(q, k, v, H, key_bias=None, scale=None, clamp=0.0):
    B, Nq, HD = q.shape
    D = HD // H
    qh = q.float().reshape(B, Nq, H, D).transpose(1, 2)
    kh = k.float().reshape(B, -1, H, D).transpose(1, 2)
    vh = v.float().reshape(B, -1, H, D).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * (scale if scale is not None else D ** -0.5)
    if clamp > 0:
        s = s.clamp(-clamp, clamp)
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, Nq, HD)


def check_attention(dev, B, H, D, Nq, Nk, mask=False, clamp=0.0, nsplit=1, scale=None, big=False, seed=0, kvlen=False):
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(seed)
    amp = 4.0 if big else 1.0
    q = (torch.randn(B, Nq, H * D, generator=g) * amp).to(H16)
    k = torch.randn(B, Nk, H * D, generator=g).to(H16)
    v = torch.randn(B, Nk, H * D, generator=g).to(H16)
    kb = kl = None
    if mask:
        kb = torch.zeros(B, Nk)
        kl = torch.zeros(B, dtype=torch.int32)
        for b in range(B):
            kb[b, max(1, Nk // (b + 2)):] = -1e30
            kl[b] = max(1, Nk // (b + 2))
    ref = _ref_attention(q, k, v, H, kb, scale, clamp)
    pad = (-Nk) % 8
    vt = F.pad(v, (0, 0, 0, pad)).transpose(1, 2).contiguous()
    out = ops.attention(q.to(dev), k.to(dev), vt.to(dev), H, D, key_bias=kb.to(dev) if kb is not None else None,
                        scale=scale, clamp=clamp, nsplit=nsplit, nk=Nk, kv_len=kl.to(dev) if (kvlen and kl is not None) else None)
    return _stat(f"attn D={D} H={H} Nq={Nq} Nk={Nk} mask={mask} clamp={clamp} nsplit={nsplit} kvlen={kvlen}", out, ref)


def check_attention_strided(dev):
    """q|k fused projection views (row stride 2C) as used by the BERT layers."""
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, H, D = 2, 256, 12, 64
    qk = torch.randn(B, T, 2 * H * D, generator=g).to(H16)
    v = torch.randn(B, T, H * D, generator=g).to(H16)
    ref = _ref_attention(qk[:, :, :H * D], qk[:, :, H * D:], v, H)
    qkd = qk.to(dev)
    out = ops.attention(qkd[:, :, :H * D], qkd[:, :, H * D:], v.transpose(1, 2).contiguous().to(dev), H, D)
    return _stat("attn strided q|k views (BERT)", out, ref)


def check_patch_embed(dev):
    """mq_patch_embed_fwd (csrc/patch_embed.hip): Swin PatchEmbed projection + patch_embed.norm + the first norm1 vs F.conv2d + two
    F.layer_norm on the same 16-bit-rounded pixels / weights; C = 96 and 192, widths that are / are not multiples of 16 patches."""
    from mq_det_amd import ops
    res = []
    g = torch.Generator().manual_seed(29)
    for (B, Hi, Wi, C) in ((2, 32, 64, 96), (1, 20, 76, 96), (2, 16, 132, 192)) + (() if QUICK else ((8, 800, 1344, 96),)):
        img = torch.randn(B, 3, Hi, Wi, generator=g).to(H16)
        w = (torch.randn(C, 3, 4, 4, generator=g) * 0.2).to(H16)
        bias, g0, b0, g1, b1 = [torch.randn(C, generator=g) * s_ + o_ for s_, o_ in ((0.1, 0), (0.2, 1), (0.1, 0), (0.2, 1), (0.1, 0))]
        y = F.conv2d(img.float(), w.float(), bias, stride=4).flatten(2).transpose(1, 2)
        x_ref = F.layer_norm(y, (C,), g0, b0, 1e-5)
        h_ref = F.layer_norm(x_ref, (C,), g1, b1, 1e-5)
        nhwc = img.to(dev).contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        x32, h1 = ops.patch_embed(nhwc, ops.patch_embed_pack(w.float()).to(H16).to(dev), *[t.to(dev) for t in (bias, g0, b0, g1, b1)])
        res.append(_stat(f"patch_embed B={B} {Hi}x{Wi} C={C}: fp32 stream LN_0(proj)", x32, x_ref, tol=1e-3))
        res.append(_stat(f"patch_embed B={B} {Hi}x{Wi} C={C}: norm1 output", h1, h_ref, tol=2e-3))
        # the caller's fp32 NCHW pixels, rounded in the kernel: the same numbers (the values above are 16-bit-representable already)
        x32b, h1b = ops.patch_embed(img.float().to(dev).contiguous(), ops.patch_embed_pack(w.float(), nchw=True).to(H16).to(dev),
                                    *[t.to(dev) for t in (bias, g0, b0, g1, b1)])
        res.append(_stat(f"patch_embed B={B} {Hi}x{Wi} C={C}: fp32 NCHW pixels == channels-last 16-bit pixels (stream)", x32b, x32.float().cpu(), tol=2e-6))
        res.append(_stat(f"patch_embed B={B} {Hi}x{Wi} C={C}: fp32 NCHW pixels == channels-last 16-bit pixels (norm1: one 16-bit ulp, other k order)", h1b, h1.float().cpu(), tol=1e-3))
    return res


def check_attention_text(dev):
    """mq_attn_text_fwd (csrc/attn_text.hip): q | k | v slices of ONE [B, T, 3 H D] projection, V row-major and transposed out of LDS, the
    register / LDS variant picked by the host bound max_kv.  Caption lengths on both sides of the 160-key variant switch and of 16-key
    block edges, per-item kv_len below the bound, the clamped variant with logits at the clamp, D = 64 (BERT) and 32, T < 256."""
    from mq_det_amd import ops
    res = []
    g = torch.Generator().manual_seed(23)
    cases = [(2, 12, 64, 256, 141, 0.0, 1.0), (2, 12, 64, 256, 160, 50000.0, 1.0), (1, 12, 64, 256, 161, 0.0, 1.0), (2, 12, 64, 256, 256, 0.0, 1.0),
             (2, 8, 32, 200, 77, 0.0, 1.0), (1, 4, 64, 40, 17, 3.0, 6.0), (3, 12, 64, 256, 16, 0.0, 1.0)]
    if QUICK:
        cases = [cases[0], cases[2], cases[4], cases[5]]
    for (B, H, D, T, kv, clamp, amp) in cases:
        qkv = (torch.randn(B, T, 3 * H * D, generator=g) * amp).to(H16)
        kl = torch.tensor([max(1, kv - 9 * b) for b in range(B)], dtype=torch.int32)            # item b has a shorter caption
        kb = torch.zeros(B, T)
        for b in range(B):
            kb[b, int(kl[b]):] = -1e30
        HD = H * D
        ref = _ref_attention(qkv[..., :HD], qkv[..., HD:2 * HD], qkv[..., 2 * HD:], H, kb, None, clamp)
        for mk in sorted({kv, 0, min(T, kv + 20)}):
            if not ops.attention_text_fits(T, kl, mk):
                continue          # precise mode on the device: up to 160 live keys (longer captions run mq_attn_resident_fwd, pipeline.bert_layer)
            out = ops.attention_text(qkv.to(dev), H, key_bias=kb.to(dev), clamp=clamp, kv_len=kl.to(dev), max_kv=mk)
            res.append(_stat(f"attn_text B={B} H={H} D={D} T={T} kv={kv} max_kv={mk} clamp={clamp}", out, ref))
        if ops.attention_text_fits(T):
            out = ops.attention_text(qkv.to(dev), H, key_bias=kb.to(dev), clamp=clamp)            # no kv_len: every block visited, bias masks
            res.append(_stat(f"attn_text B={B} H={H} D={D} T={T} kv={kv} no kv_len clamp={clamp}", out, ref))
    return res


def check_gcp_attn_fused(dev):
    """mq_gcp_attn_fwd (csrc/gcp_fused.hip: LayerNorm, to_q, sparse gather-attention, to_out, gate MLP, gated residual, next LayerNorm in one
    launch) against the unfused chain stated in plain torch with the same rounding points (tests/ops_emulation.gcp_attention).  Rows without any
    vision query (sup == 0 exactly: x_out == x bit for bit), partly padded slots, M not a multiple of the row block, both row-block sizes, with
    and without the trailing LayerNorm / the gate output."""
    import ops_emulation as emu
    from mq_det_amd import ops
    res = []
    g = torch.Generator().manual_seed(41)
    for (B, T, V, S, rb, lnf) in ((2, 144, 200, 5, 16, True), (3, 23, 37, 8, 32, True), (1, 256, 200, 5, 32, False), (8, 96, 120, 5, 0, True))[:3 if QUICK else 4]:
        if ops.f32_operands() and rb == 32:
            rb = 16                                                       # the fp32-operand build holds 16 rows per workgroup
        x = torch.randn(B, T, 768, generator=g) * 1.5
        kv = torch.randn(B, V, 1024, generator=g).to(H16)
        idx = torch.randint(0, V, (B, T, S), generator=g).to(torch.int32)
        idx[torch.rand(B, T, S, generator=g) < 0.3] = -1                 # padded slots
        idx[:, ::7] = -1                                                  # tokens without a vision query
        wq = (torch.randn(512, 768, generator=g) / 768 ** 0.5).to(H16)
        wout = (torch.randn(768, 512, generator=g) / 512 ** 0.5).to(H16)
        wg1 = (torch.randn(384, 768, generator=g) / 768 ** 0.5).to(H16)
        w2 = (torch.randn(384, generator=g) * 0.1).to(H16)
        lns = [((torch.rand(768, generator=g) + 0.5).to(H16), (torch.randn(768, generator=g) * 0.1).to(H16)) for _ in range(3)]
        ref = emu.gcp_attention(x, kv, idx, wq, wout, wg1, w2, lns[0], lns[1], lns[2] if lnf else None, want_gate=True)
        d = lambda t_: t_.to(dev)                                        # noqa: E731
        got = ops.gcp_attention(d(x), d(kv), d(idx), d(wq), d(wout), d(wg1), d(w2), tuple(map(d, lns[0])), tuple(map(d, lns[1])),
                                tuple(map(d, lns[2])) if lnf else None, want_gate=True, rows_per_block=rb)
        tag = f"gcp attention half fused B={B} T={T} V={V} S={S} rows/block={rb or 'auto'}"
        # five 16-bit rounding points in a row (LN, q, att, sup, LN, h): where the fused kernel's fp32 sums differ from torch's in the last bit
        # an intermediate flips by one 16-bit ulp -- a COMPOUND row (elem_gate off, like the block checks); with fp32 operands: zero violations
        res.append(_stat(f"{tag}: x_out (fp32 stream)", got[0], ref[0], tol=2e-3, elem_gate=False))
        if lnf:
            res.append(_stat(f"{tag}: LN_f(x_out)", got[1], ref[1], tol=2e-3, elem_gate=False))
        res.append(_stat(f"{tag}: gate", got[-1], ref[-1], tol=4e-3, elem_gate=False))
        dead = (idx < 0).all(-1)
        same = torch.equal(got[0].cpu()[dead], x[dead])
        res.append({"name": f"{tag}: rows without a vision query pass through bit for bit", "max_err": 0.0 if same else 1.0, "mean_err": 0.0, "ref_absmax": 1.0,
                    "norm_err": 0.0 if same else 1.0, "tol": 0.0, "ok": bool(same)})
    return res


def check_bert_attn_qkv(dev):
    """mq_bert_attn_qkv_fwd (csrc/bert_attn.hip): q | k | v projection + attention of every (batch item, head) in one launch, against the plain
    statement  qkv = round16(x W^T + b);  softmax(clamp(scale q k^T) + mask) v  on the same rounded operands.  Token counts on both sides of the
    160-token variant switch and of 16-token block edges (the compacted text is 16 ceil(caption / 16) long, but any T <= 256 is legal), per-item
    kv_len below T, T not a multiple of 16, the clamped variant with logits AT the clamp, a hidden-state view with padded row stride."""
    from mq_det_amd import ops
    res = []
    g = torch.Generator().manual_seed(31)
    C, H = 768, 12
    cases = [(2, 144, 141, 0.0, 1.0), (2, 160, 160, 50000.0, 1.0), (1, 176, 161, 0.0, 1.0), (2, 256, 256, 0.0, 1.0), (3, 40, 17, 3.0, 8.0),
             (2, 23, 23, 0.0, 1.0), (8, 96, 81, 50000.0, 1.0)]
    if QUICK:
        cases = [cases[0], cases[2], cases[4], cases[5]]
    for (B, T, kv, clamp, amp) in cases:
        if not ops.bert_attention_qkv_fits(T, C, H):
            continue              # precise mode on the device: up to 160 tokens
        xs = torch.randn(B, T, C + 8, generator=g).to(H16)
        x = xs[:, :, :C]                                                   # row stride C + 8: a view, as the LayerNorm kernel may hand it over
        w = torch.randn(3 * C, C, generator=g) / math.sqrt(C)
        w[:2 * C] *= amp                                                   # (q and k only: logits that reach the clamp, values of unit size)
        w = w.to(H16)
        bq = (torch.randn(3 * C, generator=g) * 0.1).to(H16)
        kl = torch.tensor([max(1, kv - 9 * b) for b in range(B)], dtype=torch.int32)
        kb = torch.zeros(B, T)
        for b in range(B):
            kb[b, int(kl[b]):] = -1e30
        qkv = F.linear(x.float(), w.float(), bq.float()).to(H16)
        ref = _ref_attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], H, kb, None, clamp)
        xd = xs.to(dev)[:, :, :C]
        for use_len in (True, False):
            out = ops.bert_attention_qkv(xd, w.to(dev), bq.to(dev), H, key_bias=kb.to(dev), clamp=clamp, kv_len=kl.to(dev) if use_len else None)
            res.append(_stat(f"attn bert qkv fused B={B} T={T} kv={kv} clamp={clamp} kv_len={'yes' if use_len else 'no'}", out, ref))
    return res


def _tiny(dev, image_hw=(160, 192), B=2, seed=0, spec=None):
    """Shared tiny model: oracle state dict + product model with the same weights (spec: tiny_spec() = Swin-T widths, or
    tiny_l_spec() = Swin-L widths / window 12)."""
    from oracle import tiny_spec
    from oracle.weights import make_state_dict
    from mq_det_amd import get_cfg
    from mq_det_amd.modeling.detector import GeneralizedVLRCNN_New
    spec = spec or tiny_spec()
    sd = make_state_dict(spec, seed)
    cfg = get_cfg()
    cfg.MODEL.SWINT.EMBED_DIM, cfg.MODEL.SWINT.NUM_HEADS = spec.swin_embed, spec.swin_heads
    cfg.MODEL.SWINT.WINDOW_SIZE, cfg.MODEL.SWINT.OUT_CHANNELS = spec.window, spec.swin_dims
    cfg.MODEL.SWINT.DEPTHS = spec.swin_depths
    cfg.MODEL.LANGUAGE_BACKBONE.NUM_HIDDEN_LAYERS = spec.bert_layers
    cfg.MODEL.LANGUAGE_BACKBONE.QV_START = spec.qv_start
    cfg.MODEL.LANGUAGE_BACKBONE.BERT_VOCAB_SIZE = spec.vocab
    cfg.MODEL.DYHEAD.NUM_CONVS = spec.dyhead_convs
    cfg.MODEL.DYHEAD.NUM_CLASSES = spec.num_classes
    cfg.MODEL.ATSS.DETECTIONS_PER_IMG = spec.detections_per_img
    cfg.MODEL.COMPUTE_DTYPE = _DTYPE_NAME[H16]
    model = GeneralizedVLRCNN_New(cfg, tokenizer=object())
    model.load_state_dict(sd, strict=True)
    model.to(dev)
    P = model.prepare(dev)
    return spec, sd, cfg, model, P


_CACHE = {}
_DTYPE_NAME = {torch.float16: "float16", torch.bfloat16: "bfloat16", torch.float32: "float32"}


def tiny(dev, large=False):
    key = "tiny_l" if large else "tiny"
    if key not in _CACHE:
        from oracle import tiny_l_spec
        _CACHE[key] = _tiny(dev, spec=tiny_l_spec() if large else None)
    return _CACHE[key]


def check_swin_fpn(dev, large=False):
    """large: Swin-L widths (192 .. 1536), heads 6 .. 48, window 12 (144-token windows) -- MQ-GLIP-L's backbone."""
    from oracle import backbone as ob
    from mq_det_amd.modeling import pipeline
    spec, sd, cfg, model, P = tiny(dev, large)
    img = torch.randn(2, 3, 90, 122, generator=torch.Generator().manual_seed(1)).to(H16)
    with torch.no_grad():
        c = ob.swin_forward(sd, "backbone.body", img.float(), spec)
        p = ob.fpn_forward(sd, "backbone.fpn", c)
        x = img.to(dev).contiguous(memory_format=torch.channels_last)
        cg = pipeline.swin_forward(P, cfg, x)
        pg = pipeline.fpn_forward(P, cg)
    tag = "swin-L " if large else ""
    out = [_stat(f"{tag}swin c{i + 3}", cg[i].permute(0, 3, 1, 2), c[i + 1], tol=1e-2) for i in range(3)]
    out += [_stat(f"{tag}fpn p{i + 3}", pg[i], p[i], tol=1e-2) for i in range(5)]
    return out


def check_window_attention(dev, large=False):
    """One Swin block's attention (shifted, H/W not multiples of the window) vs the oracle's window_attention path.
    large: window 12 (144 tokens per window, the 160-padded kernel variant), Swin-L head counts."""
    from oracle import backbone as ob
    from mq_det_amd import ops
    spec, sd, cfg, model, P = tiny(dev, large)
    res = []
    # (stage 1 at 29 x 43: 70 windows -- several trips of the persistent workgroups of mq_window_attn_qkv_fwd and idle waves in the last)
    for stage, (H, W) in (((0, (29, 41)), (1, (12, 16)), (3, (5, 7))) if large else ((0, (23, 31)), (1, (12, 16)), (1, (29, 43)), (0, (40, 57)))):
        C, heads, ws = spec.swin_dims[stage], spec.swin_heads[stage], spec.window
        for shift in (0, ws // 2):
            b = f"backbone.body.layers.{stage}.blocks.{1 if shift else 0}.attn"
            g = torch.Generator().manual_seed(7 + shift)
            y = torch.randn(2, H, W, C, generator=g).to(H16)            # = norm1(x)
            # oracle: pad, roll, partition, attention (without proj), reverse, roll back, crop
            yf = y.float()
            pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
            yp = F.pad(yf, (0, 0, 0, pad_r, 0, pad_b))
            Hp, Wp = H + pad_b, W + pad_r
            if shift:
                yp = torch.roll(yp, (-shift, -shift), (1, 2))
            xw = ob.to_windows(yp, ws)
            qkv = F.linear(xw, sd[b + ".qkv.weight"].to(H16).float(), sd[b + ".qkv.bias"].to(H16).float())
            Bw, N, _ = qkv.shape
            qkv = qkv.to(H16).float().reshape(Bw, N, 3, heads, 32).permute(2, 0, 3, 1, 4)
            attn = (qkv[0] * 32 ** -0.5) @ qkv[1].transpose(-1, -2)
            bias = sd[b + ".relative_position_bias_table"][ob.rel_pos_index(ws).reshape(-1)].reshape(N, N, heads).permute(2, 0, 1)
            attn = attn + bias[None]
            if shift:
                m = ob.shift_mask(Hp, Wp, ws, shift)
                nW = m.shape[0]
                attn = (attn.reshape(Bw // nW, nW, heads, N, N) + m[None, :, None]).reshape(Bw, heads, N, N)
            o = (attn.softmax(-1) @ qkv[2]).transpose(1, 2).reshape(Bw, N, C)
            o = ob.from_windows(o, ws, 2, Hp, Wp)
            if shift:
                o = torch.roll(o, (shift, shift), (1, 2))
            ref = o[:, :H, :W]
            qkv_dev = F.linear(y.to(dev), P[b + ".qkv.weight"], P[b + ".qkv.bias"]).reshape(2, H, W, 3 * C)
            got = ops.window_attention(qkv_dev, P[b + ".qkv.bias"], P[b + ".rel_bias"], heads, ws, shift)
            res.append(_stat(f"window_attn ws={ws} stage{stage} {H}x{W} shift={shift}", got, ref, tol=1.5e-3))
            if C in ops.WINDOW_QKV_WIDTHS and ws * ws <= 64:            # the qkv projection inside the kernel (mq_window_attn_qkv_fwd)
                got2 = ops.window_attention_qkv(y.to(dev), P[b + ".qkv.weight"], P[b + ".qkv.bias"], P[b + ".rel_bias"], heads, ws, shift)
                res.append(_stat(f"window_attn with the qkv projection inside ws={ws} stage{stage} {H}x{W} shift={shift}", got2, ref, tol=1.5e-3))
                res.append(_stat("... vs GEMM + mq_window_attn_fwd", got2, got.float().cpu(), tol=2e-3))
    return res


def _gcp_inputs(spec, T=40):
    g = torch.Generator().manual_seed(11)
    B, C = 2, spec.bert_hidden
    x = torch.randn(B, T, C, generator=g).to(H16)
    vis = torch.randn(B, 15, C, generator=g).to(H16)
    vis[1, 10:] = 0
    tok = {0: [2], 1: [5, 6, 7], 2: [10, 11]}
    vmask = torch.zeros(B, 15, T)
    idx = torch.full((B, T, 5), -1, dtype=torch.int32)
    for lab, toks in tok.items():
        for b in range(B):
            if b == 1 and lab == 2:
                continue
            vmask[b, lab * 5:(lab + 1) * 5, toks] = 1
            for t in toks:
                idx[b, t] = torch.arange(lab * 5, lab * 5 + 5, dtype=torch.int32)
    return x, vis, vmask, idx


def check_gcp_block(dev):
    from oracle import language as ol
    from mq_det_amd.modeling import pipeline
    spec, sd, cfg, model, P = tiny(dev)
    x, vis, vmask, idx = _gcp_inputs(spec)
    b = "language_backbone.body.model.encoder.qv_layer.0"
    with torch.no_grad():
        ref = ol.gated_cross_attention_block(sd, b, x.float(), vis.float(), vmask, spec)
        got = pipeline.gcp_block(P, b, x.to(dev), vis.to(dev), idx.to(dev))
        # exactly-zero cross attention for tokens without queries (quirk 5)
        from mq_det_amd import ops
        q = F.linear(pipeline._ln(P, b + ".attn.norm", x.to(dev)), P[b + ".attn.to_q.weight"])
        kv = F.linear(pipeline._ln(P, b + ".attn.norm_kv", vis.to(dev)), P[b + ".attn.to_kv.weight"])
        sup = ops.gcp_sparse_attention(q, kv, idx.to(dev))
    noq = (vmask.sum(1) == 0)
    z = _stat("gcp sparse attn: zero rows for tokens w/o query", sup.cpu()[noq], torch.zeros_like(sup.cpu()[noq]), tol=0.0)
    return [_stat("gcp gated cross-attention block", got, ref, tol=2.5e-3), z]


def check_pre_select(dev):
    from oracle import language as ol
    from mq_det_amd.modeling import pipeline
    spec, sd, cfg, model, P = tiny(dev)
    g = torch.Generator().manual_seed(12)
    vis = torch.randn(2, 15, spec.fpn_out, generator=g).to(H16)
    img = torch.randn(2, 333, spec.fpn_out, generator=g).to(H16)
    p = "language_backbone.body.model.pre_select"
    with torch.no_grad():
        ref = ol.pre_select(sd, p, vis.float(), img.float(), spec)
        got = pipeline.pre_select(P, p, vis.to(dev), img.to(dev), 1.0)
    return _stat("gcp pre-select (2 layers, Nk=333 ragged)", got, ref, tol=2.5e-3)


def check_bert_layer(dev, clamp):
    from oracle import language as ol
    from mq_det_amd.modeling import pipeline
    spec, sd, cfg, model, P = tiny(dev)
    g = torch.Generator().manual_seed(13)
    T = 64
    x = torch.randn(2, T, spec.bert_hidden, generator=g).to(H16)
    am = torch.ones(2, T, dtype=torch.long)
    am[0, 25:] = 0
    am[1, 50:] = 0
    b = "rpn.head.dyhead_tower.1" if clamp else "language_backbone.body.model.encoder.layer.0"
    with torch.no_grad():
        ref = ol.bert_layer(sd, b, x.float(), ol.extended_mask(am), spec.bert_heads, spec.bert_eps, clamp=clamp)
        kb = ((1.0 - am.float()) * -1e30).to(dev)
        got = pipeline.bert_layer(P, b, x.to(dev), kb, clamp)
    return _stat(f"bert layer clamp={clamp}", got, ref, tol=2.5e-3)


def check_bert_clamp_fused(dev):
    """KERNELS["BERT_CLAMP_FUSED"]: mq_clamp_gelu_clamp and mq_layernorm_clamp_fwd against the torch passes they replace -- on values that DO
    reach the clamp (the +-50000 of the reference and a small one), fp16 / fp32 input x residual, with and without the fp32 output;
    then the clamped BERT layer of the tiny model with the switch on against the oracle and against the switch off."""
    from oracle import language as ol
    from mq_det_amd import ops
    from mq_det_amd.modeling import pipeline
    spec, sd, cfg, model, P = tiny(dev)
    g = torch.Generator().manual_seed(37)
    res = []
    for cl, amp in ((50000.0, 60000.0), (2.5, 4.0), (50000.0, 1.0)):
        x = ((torch.rand(3, 40, 768, generator=g) * 2 - 1) * amp).to(H16)        # finite in fp16 (|x| <= 60000), a sixth beyond the clamp
        ref = F.gelu(x.float().clamp(-cl, cl)).to(H16).clamp(-cl, cl)
        res.append(_stat(f"clamp_gelu_clamp clamp={cl} |x|~{amp}", ops.clamp_gelu_clamp(x.to(dev), cl), ref.float(), tol=1e-3))
        gam, bet = (torch.randn(768, generator=g) * 0.2 + 1).to(H16), (torch.randn(768, generator=g) * 0.1).to(H16)
        if amp > 10:
            gam, bet = gam * 25000, bet * 25000                         # LayerNorm outputs beyond the clamp as well
        for xf in (False, True):
            for rf in (False, True):
                xx = (x.float() if xf else x).to(dev)
                rr = torch.randn(3, 40, 768, generator=g) * min(amp, 100.0)
                rr = (rr if rf else rr.to(H16)).to(dev)
                a16, a32 = ops.layer_norm(xx, gam.to(dev), bet.to(dev), 1e-12, residual=rr, want_sum=False, want_y32=True, clamp=cl)
                b16, b32 = ops.layer_norm(xx.clamp(-cl, cl), gam.to(dev), bet.to(dev), 1e-12, residual=rr, want_sum=False, want_y32=True)
                # (equal bit for bit through the emulation; the bounds leave room for one rounding step should the two LayerNorm kernels
                # contract a multiply-add differently on the device, where this pair has not run side by side yet)
                res.append(_stat(f"layer_norm clamp={cl} |x|~{amp} x fp32={xf} res fp32={rf}: y vs clamp -> LayerNorm -> clamp", a16, b16.clamp(-cl, cl).float().cpu(), tol=1e-3))
                res.append(_stat(f"layer_norm clamp={cl} |x|~{amp} x fp32={xf} res fp32={rf}: y32 vs clamp -> LayerNorm -> clamp", a32, b32.clamp(-cl, cl).float().cpu(), tol=2e-6))
    T = 64
    x = torch.randn(2, T, spec.bert_hidden, generator=g).to(H16)
    am = torch.ones(2, T, dtype=torch.long)
    am[0, 25:] = 0
    b = "rpn.head.dyhead_tower.1"
    saved = ops.KERNELS["BERT_CLAMP_FUSED"]
    try:
        with torch.no_grad():
            ref = ol.bert_layer(sd, b, x.float(), ol.extended_mask(am), spec.bert_heads, spec.bert_eps, clamp=True)
            kb = ((1.0 - am.float()) * -1e30).to(dev)
            outs = {}
            for mode in (0, 1):
                ops.KERNELS["BERT_CLAMP_FUSED"] = mode
                outs[mode] = pipeline.bert_layer(P, b, x.to(dev), kb, True)
        res.append(_stat("bert layer clamp=True, clamps inside the kernels: vs the oracle", outs[1], ref, tol=2.5e-3))
        res.append(_stat("bert layer clamp=True, clamps inside the kernels: vs the torch passes", outs[1], outs[0].float().cpu(), tol=1e-3))
    finally:
        ops.KERNELS["BERT_CLAMP_FUSED"] = saved
    return res


def check_vl_fuse(dev):
    from oracle import head as oh
    from mq_det_amd.modeling import pipeline
    spec, sd, cfg, model, P = tiny(dev)
    g = torch.Generator().manual_seed(14)
    sizes = [(20, 24), (10, 12), (5, 6), (3, 3), (2, 2)]
    feats = [torch.randn(2, 256, h, w, generator=g).to(H16) for h, w in sizes]
    l = torch.randn(2, 64, spec.bert_hidden, generator=g).to(H16)
    am = torch.ones(2, 64, dtype=torch.long)
    am[0, 30:] = 0
    b = "rpn.head.dyhead_tower.0.b_attn"
    with torch.no_grad():
        rv, rl = oh.vl_fuse(sd, b, [f.float() for f in feats], l.float(), am, spec)
        kb = ((1.0 - am.float()) * -1e30).to(dev)
        gv, gl = pipeline.vl_fuse(P, b, [f.to(dev).contiguous(memory_format=torch.channels_last) for f in feats], l.to(dev), kb)
    out = [_stat(f"vlfuse image side lvl{i}", gv[i], rv[i], tol=2.5e-3) for i in range(5)]
    out.append(_stat("vlfuse text side", gl, rl, tol=2.5e-3))
    return out


def check_fusion_layer(dev, sizes=((100, 168), (50, 84), (25, 42), (13, 21), (7, 11)), n_tok=141, B=1):
    """ONE whole fusion layer of the head -- VLFuse (both directions), the clamped BERT layer, DyConv (offset convs, 13 DCNv2 branches,
    GroupNorm, scale attention, DyReLU) -- at the FULL geometry of an 800 x 1333 image (N = 22 400 pyramid tokens, T = 256 with n_tok live
    text tokens) against the oracle's layer on the same inputs.  Default sizes: every tile / level boundary of the benchmark shape."""
    from oracle import head as oh, language as ol
    from mq_det_amd.modeling import pipeline
    spec, sd, cfg, model, P = tiny(dev)
    g = torch.Generator().manual_seed(33)
    T = spec.max_query_len
    feats = [torch.randn(B, 256, h, w, generator=g).to(H16) for h, w in sizes]
    l = torch.randn(B, T, spec.bert_hidden, generator=g).to(H16)
    am = torch.zeros(B, T, dtype=torch.long)
    am[:, :n_tok] = 1
    t = "rpn.head.dyhead_tower"
    with torch.no_grad():
        rv, rl = oh.vl_fuse(sd, f"{t}.0.b_attn", [f.float() for f in feats], l.float(), am, spec)
        rl2 = ol.bert_layer(sd, f"{t}.1", rl, ol.extended_mask(am), spec.bert_heads, clamp=True)
        rd = oh.dyconv(sd, f"{t}.2", rv, spec)
        kb = ((1.0 - am.float()) * -1e30).to(dev)
        kv_len = torch.full((B,), n_tok, dtype=torch.int32, device=dev)
        gv, gl = pipeline.vl_fuse(P, f"{t}.0.b_attn", [f.to(dev).contiguous(memory_format=torch.channels_last) for f in feats], l.to(dev), kb,
                                  kv_len=kv_len, max_kv=n_tok)
        gl16, gl32 = (gl.to(H16), gl.float()) if P["_r32"] else (gl, None)
        gl2 = pipeline._bert(P, f"{t}.1", gl16, gl32, kb, True, kv_len)
        gl2 = gl2[1] if gl2[1] is not None else gl2[0]
        gd = pipeline.dyconv(P, cfg, f"{t}.2", gv)
    live = am.bool()
    out = [_stat(f"fusion layer @ {sizes[0][0]}x{sizes[0][1]}: image tokens after VLFuse lvl{i}", gv[i], rv[i], tol=8e-3) for i in range(len(sizes))]
    out.append(_stat("fusion layer: text hidden after VLFuse (live tokens)", gl.cpu()[live], rl[live], tol=8e-3))
    out.append(_stat("fusion layer: text hidden after the clamped BERT layer (live tokens)", gl2.cpu()[live], rl2[live], tol=1e-2))
    out += [_stat(f"fusion layer: image tokens after DyConv lvl{i}", gd[i], rd[i], tol=1.5e-2) for i in range(len(sizes))]
    return out


def check_vlfuse_kernels(dev):
    """The two VLFuse attention kernels (vlfuse_attn.hip) against a plain fp32 restatement on the same fp16 inputs:
    every register-tile variant (NT = 1..4 text tiles), ragged N / T, kv_len, masked-bias keys, several key splits.
    Inputs are random (not symmetric), so a transposed operand or a wrong k-slot permutation cannot cancel out."""
    import ops_emulation as emu
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(21)
    res = []
    # (141 tokens: NT = 3 tiles with ONE live 16-key block in the last -- Q fragments in registers by default; kv_len = 1: a batch item
    # with a single live key; 83 / 110 / 177 keys: 2 / 3 / 4 live blocks in the last tile, staged only as far as they are read)
    for B, N, T, kv in ((2, 645, 64, None), (3, 300, 100, [100, 37, 70]), (2, 200, 160, [131, 160]), (1, 130, 256, [256]),
                        (2, 100, 141, [141, 1]), (1, 70, 96, [83]), (1, 70, 110, [110]), (1, 70, 200, [177]),
                        (9, 128, 40, None))[1 if QUICK else 0:6 if QUICK else 9]:
        v_ln = torch.randn(B, N, 256, generator=g).to(H16)
        kf = (torch.randn(B, 8, T, 256, generator=g) / 8).to(H16)
        vo = torch.randn(B, 8, T, 256, generator=g).to(H16)
        bias = torch.randn(B, 8, T, generator=g)
        bias[:, :, T // 3] = -1e30                       # a masked key in the middle of the valid range
        ob = torch.randn(256, generator=g).to(H16)
        kv_len = None if kv is None else torch.tensor(kv, dtype=torch.int32)
        ref = emu.vlfuse_i2t(v_ln.float(), kf.float(), vo.float(), bias, ob.float(), kv_len, 0)
        for variant in ((0,) if ops.f32_operands() == 1 else (0, 1)):          # (precise mode on the device: one variant, see ops.vlfuse_i2t)
            got = ops.vlfuse_i2t(v_ln.to(dev), kf.to(dev), vo.to(dev), bias.to(dev), ob.to(dev),
                                 None if kv_len is None else kv_len.to(dev), max_kv=0 if kv is None else max(kv), variant=variant)
            res.append(_stat(f"vlfuse image side [{'Q in registers where it fits' if variant == 0 else 'Q tile in LDS beyond 128 keys'}] B={B} N={N} T={T} kv_len={kv}", got, ref, tol=2e-3))
    for B, N, T, ns, kv in ((2, 645, 64, 1, None), (1, 22400, 256, 6, None), (3, 1000, 100, 3, None), (2, 130, 160, 2, [160, 90]),
                            (9, 70, 40, 1, None), (3, 500, 256, 4, [256, 128, 77]), (2, 300, 256, 3, [141, 1]),
                            (3, 200, 256, 2, [17, 141, 96]))[3 if QUICK else 0:]:
        v_ln = torch.randn(B, N, 256, generator=g).to(H16)
        kf = (torch.randn(B, 8, T, 256, generator=g) / 8).to(H16)
        kv_len = None if kv is None else torch.tensor(kv, dtype=torch.int32)
        ref = emu.vlfuse_t2i(kf.float(), v_ln.float(), ns, kv_len=kv_len)
        # max_kv sizes the grid (live (head, 16-row block) units packed over the waves); without it the grid covers all T rows
        for mk in ((0,) if kv is None else (max(kv), 0)):
            got = ops.vlfuse_t2i(kf.to(dev), v_ln.to(dev), ns, kv_len=None if kv is None else kv_len.to(dev), max_kv=mk)
            res.append(_stat(f"vlfuse text side B={B} N={N} T={T} nsplit={ns} kv_len={kv} max_kv={mk}", got, ref, tol=2e-3))
    return res


def check_dcn(dev):
    """Fused DCNv2 kernel (single and grouped launch) vs the oracle's dcn_v2, incl. stride 2 and the flat-offset-indexing
    quirk (offsets from a bigger level), plus the GroupNorm statistics of its epilogue."""
    from oracle import head as oh
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(15)
    res = []
    group, refs = [], []
    for name, (H, W), (oH, oW), stride in (("same level s1", (13, 17), (13, 17), 1), ("from level-1 s2", (26, 33), (13, 17), 2),
                                           ("from level+1 (quirk)", (7, 9), (13, 17), 1)):
        x = torch.randn(2, 256, H, W, generator=g).to(H16)
        om = torch.randn(2, 27, oH, oW, generator=g) * 1.5
        w = (torch.randn(256, 256, 3, 3, generator=g) / 48).to(H16)
        bias = torch.randn(256, generator=g).to(H16)
        ref = oh.dcn_v2(x.float(), om[:, :18], om[:, 18:].sigmoid(), w.float(), bias.float(), stride)
        wp = w.permute(0, 2, 3, 1).reshape(256, -1).to(dev)
        Ho, Wo = ref.shape[-2:]
        wy = torch.rand(Ho, generator=g).to(dev)
        wx = torch.rand(Wo, generator=g).to(dev)
        xd, omd = x.to(dev).permute(0, 2, 3, 1).contiguous(), om.to(dev).contiguous()
        y2, _, sums = ops.dcnv2(xd, omd, wp, bias.to(dev), stride, want_stats=True, wy=wy, wx=wx)
        res.append(_stat(f"dcnv2 fused implicit-GEMM {name}", y2.reshape(2, Ho, Wo, 256).permute(0, 3, 1, 2), ref, tol=4e-3))
        # statistics emitted by the kernel's epilogue == the same sums over its own fp16 output
        yf = y2.float()
        wpos = (wy[:, None] * wx[None, :]).reshape(-1)
        st_ref = torch.stack([yf.sum(1), (yf * yf).sum(1), (yf * wpos[None, :, None]).sum(1)], -1)
        res.append(_stat(f"dcnv2 fused GroupNorm statistics {name}", sums.sum(1), st_ref, tol=1e-4))
        group.append({"x": xd, "om": omd, "w": wp, "bias": bias.to(dev), "stride": stride, "wy": wy, "wx": wx})
        refs.append((name, y2, sums))
    grouped = ops.dcnv2_group(group)                                             # ONE launch for the three calls
    for (name, y2, sums), (yg, _, sg) in zip(refs, grouped):
        res.append(_stat(f"dcnv2 grouped launch == single launch: {name}", yg, y2.float(), tol=0.0))
        res.append(_stat(f"dcnv2 grouped launch statistics: {name}", sg, sums, tol=0.0))
    # GroupNorm / scale-attention coefficients: grouped launch (4 parallel reducers) vs the single-branch kernel
    gam, bet = (torch.randn(256, generator=g) * 0.1 + 1).to(H16).to(dev), (torch.randn(256, generator=g) * 0.1).to(H16).to(dev)
    aw, ab = (torch.randn(256, generator=g) * 0.1).to(dev), torch.randn(1, generator=g).to(dev)
    items = [{"sums": sg, "n": yg.shape[1], "gamma": gam, "beta": bet, "nbranches": 3} for (yg, _, sg) in grouped]
    for (name, _, _), (yg, (Ho, Wo), sg), cg in zip(refs, grouped, ops.dyconv_coef_group(items, aw, ab, 16, 1e-5)):
        c1 = ops.dyconv_branch_coef(yg, Wo, gam, bet, aw, ab, 16, 1e-5, 3, sums=sg)
        res.append(_stat(f"dyconv coefficients grouped vs single launch: {name}", cg, c1, tol=1e-5))
    return res


def check_ref_pins(dev):
    """PINS against the reference's OWN device code (oracle/_ref/libmqdet_ref.so = deform_conv_kernel_cuda.cu:474-504,577-640
    and ml_nms.cu:13-75 compiled with hipcc by oracle/build_ref.py): (1) the oracle restatements oracle.head.dcn_v2 and
    oracle.postprocess.ml_nms, (2) the HIP kernels mq_dcnv2_fwd and mq_ml_nms."""
    from oracle import head as oh, postprocess as op, ref_native as rn
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(31)
    res = []
    for name, (H, W), (oH, oW), stride, amp in (("same level s1", (13, 17), (13, 17), 1, 1.5), ("from level-1 s2", (26, 33), (13, 17), 2, 1.5),
                                                ("from level+1 (quirk 1)", (7, 9), (13, 17), 1, 1.5), ("large offsets", (20, 24), (20, 24), 1, 6.0),
                                                ("P3-like 50x84", (50, 84), (50, 84), 1, 0.7)):
        x = torch.randn(2, 256, H, W, generator=g).to(H16)
        om = torch.randn(2, 27, oH, oW, generator=g) * amp
        w = (torch.randn(256, 256, 3, 3, generator=g) / 48).to(H16)
        bias = torch.randn(256, generator=g).to(H16)
        off, msk = om[:, :18].contiguous(), om[:, 18:].sigmoid().contiguous()
        pin = rn.dcn_v2(x.float().to(dev), off.to(dev), msk.to(dev), w.float().to(dev), bias.float().to(dev), stride)
        ora = oh.dcn_v2(x.float(), off, msk, w.float(), bias.float(), stride)
        res.append(_stat(f"PIN oracle.dcn_v2 vs reference CUDA kernel: {name}", ora, pin, tol=2e-5))
        wp = w.permute(0, 2, 3, 1).reshape(256, -1).to(dev)
        y, (Ho, Wo) = ops.dcnv2(x.to(dev).permute(0, 2, 3, 1).contiguous(), om.to(dev).contiguous(), wp, bias.to(dev), stride)
        res.append(_stat(f"PIN mq_dcnv2_fwd vs reference CUDA kernel: {name}", y.reshape(2, Ho, Wo, 256).permute(0, 3, 1, 2), pin, tol=4e-3))
    for n in (1, 63, 64, 65, 700, 4100):
        xy = torch.rand(n, 2, generator=g) * 300
        wh = torch.rand(n, 2, generator=g) * 120 + 4
        boxes = torch.cat([xy, xy + wh], -1)
        scores = torch.rand(n, generator=g)
        labels = torch.randint(1, 6, (n,), generator=g)
        pin = rn.ml_nms(boxes.to(dev), scores.to(dev), labels.float().to(dev), 0.6)
        ora = op.ml_nms(boxes, scores, labels.float(), 0.6)
        order = torch.argsort(scores, descending=True, stable=True)
        keep = ops.ml_nms(boxes[order][None].contiguous().to(dev), labels[order][None].int().contiguous().to(dev),
                          torch.tensor([n], dtype=torch.int32, device=dev), 0.6).cpu()[0]
        hip = torch.sort(order[keep])[0]
        for nm, got in (("oracle.ml_nms", ora), ("mq_ml_nms", hip)):
            ok = got.shape == pin.shape and bool(torch.equal(got, pin))
            res.append({"name": f"PIN {nm} vs reference CUDA kernel + sweep: n={n} kept={len(pin)}", "max_err": 0.0 if ok else 1.0,
                        "mean_err": 0.0, "ref_absmax": 1.0, "norm_err": 0.0 if ok else 1.0, "tol": 0.0, "ok": ok})
    return res


def _flag(name, ok):
    return {"name": name, "max_err": 0.0 if ok else 1.0, "mean_err": 0.0, "ref_absmax": 1.0, "norm_err": 0.0 if ok else 1.0, "tol": 0.0, "ok": bool(ok)}


def check_post_fused(dev):
    """csrc/post2.hip -- mq_post_select_fwd / mq_post_sort_fwd / mq_post_finalize_fwd against their plain-torch restatements
    (tests/ops_emulation.py: stable sorts) on synthetic score maps: sparse and dense candidates, scores quantised to a few values so that
    the cut of the radix select falls INSIDE a group of equal keys (ties resolved by flat index), one value everywhere, no candidate at all,
    rows that are not 16-byte aligned; then the product post-processing with the four-launch path against the round-3 chain it replaces
    (torch.topk / argsort / gathers) on the same head outputs.  Integer / index work: EXACT (boxes: the same fp32 expression, 1e-6)."""
    import ops_emulation as emu
    from mq_det_amd import ops, get_cfg
    from mq_det_amd.modeling import pipeline
    res = []
    g = torch.Generator().manual_seed(17)
    cases = [("sparse", (2, [(9, 13), (5, 6), (3, 3)], 7, 40, 0.02, None)), ("dense", (2, [(20, 31), (9, 11)], 5, 300, 0.9, None)),
             ("dense, 5 score values", (2, [(20, 31), (10, 16)], 5, 120, 0.8, 5)), ("one value everywhere", (1, [(12, 17)], 3, 50, 1.0, 1)),
             ("no candidate", (2, [(6, 7), (2, 3)], 4, 30, 0.0, None)), ("k = everything", (1, [(4, 5)], 3, 60, 0.5, None)),
             ("two slices, ties across them", (1, [(60, 70), (9, 9)], 10, 300, 0.5, 64)), ("k = 1500", (1, [(50, 60)], 8, 1500, 0.9, None)),
             ("P3-sized level", (1, [(100, 168)], 40, 1000, 0.3, 4096))]
    if QUICK:
        cases = cases[:-1] + [("large level", (1, [(40, 56)], 10, 1000, 0.3, 512))]
    for name, (B, shapes, L, topn, dens, quant) in cases:
        ranked, reg, anchors, ks = [], [], [], []
        for (h, w) in shapes:
            hw = h * w
            v = torch.rand(B, hw, L, generator=g)
            if quant:
                v = (torch.floor(v * quant) + 1) / (quant + 1)
            cand = torch.rand(B, hw, L, generator=g) < dens
            ranked.append(torch.where(cand, v, torch.full_like(v, -1.0)).contiguous())
            reg.append((torch.randn(B, hw, 4, generator=g) * 2).contiguous())
            xy = torch.rand(hw, 2, generator=g) * 300
            anchors.append(torch.cat([xy, xy + 8 + torch.rand(hw, 2, generator=g) * 64], 1).contiguous())
            ks.append(min(topn, hw * L))
        label_ids = torch.randperm(L, generator=g).to(torch.int32) + 1
        im_wh = torch.tensor([[333.0, 250.0]] * B)
        eb, es, el, ei = emu.post_select(ranked, reg, anchors, ks, label_ids.long(), im_wh)
        gb, gs, gl, gi = [t.cpu() for t in ops.post_select([t.to(dev) for t in ranked], [t.to(dev) for t in reg], [t.to(dev) for t in anchors], ks,
                                                           label_ids.to(dev), im_wh.to(dev))]
        # every level's slots: the same candidates in the same order -- (value desc, flat index asc) = a stable descending sort
        same = torch.equal(ei, gi) and torch.equal(el.int(), gl.int()) and bool((es - gs).abs().max() <= 1.5e-7) and bool((eb - gb).abs().max() <= 1e-4)
        res.append(_flag(f"post_select [{name}] B={B} levels={shapes} L={L} k={ks}: the sorted top-k of every level, decoded", same))
        sb, ss, sl, sn = emu.post_sort(gb, gs, gl, ks)        # the SAME lists into both merges: everything behind is exact
        hb, hs, hl, hn = [t.cpu() for t in ops.post_sort(gb.to(dev), gs.to(dev), gl.to(dev), ks)]
        ok = torch.equal(ss, hs) and torch.equal(sl.int(), hl.int()) and torch.equal(sn.int(), hn.int()) and bool((sb - hb).abs().max() <= 1e-4)
        res.append(_flag(f"post_sort [{name}] tot={sum(ks)}: merge of the level lists (score desc, level, position), empty rows last, nvalid", ok))
        tot = sum(ks)
        for K, extra in ((max(1, tot // 3), 4), (tot, 0), (1, 1)):
            K2 = min(K + extra, tot)
            keep = (torch.rand(B, tot, generator=g) < 0.7).to(torch.uint8)
            fo, fc = emu.post_finalize(hb, hs, hl, keep, K, K2)
            ho, hc = [t.cpu() for t in ops.post_finalize(hb.to(dev), hs.to(dev), hl.to(dev), keep.to(dev), K, K2)]
            res.append(_flag(f"post_finalize [{name}] K={K} K2={K2}: packed rows, counts, overflow flag", torch.equal(fc.int(), hc.int()) and bool((fo - ho).abs().max() <= 1e-4)))
    return res


def check_post_golden(dev, golden_dir=None):
    """mq_align_scores_fwd / mq_box_decode / the whole product post-processing on the inputs of the reference-generated
    fixtures tests/golden/atss_post_{dyhead,mdetr}.npz (outputs of the reference's own ATSSPostProcessor): class scores and
    candidate values to 1e-5 (fp32 logits in), exact labels, boxes to 1e-3 px (bbox_reg passes through fp16 here)."""
    import os
    import numpy as np
    from oracle import postprocess as op
    from mq_det_amd import ops, get_cfg
    from mq_det_amd.modeling import pipeline
    from mq_det_amd.modeling.query_selector import build_token_index
    golden_dir = golden_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    res = []
    for name, mdetr, ndet in (("atss_post_dyhead", -1, 100), ("atss_post_mdetr", 3000, 100)):
        gd = np.load(os.path.join(golden_dir, name + ".npz"))
        t = lambda k: torch.from_numpy(gd[k])                                   # noqa: E731
        keys, lens, flat = gd["pmap_keys"], gd["pmap_lens"], gd["pmap_flat"]
        pm, o = {}, 0
        for k, n in zip(keys.tolist(), lens.tolist()):
            pm[int(k)] = [int(v) for v in flat[o:o + n]]
            o += n
        sizes = [tuple(int(v) for v in r) for r in gd["sizes"]]
        labels = [k for k, v in pm.items() if len(v)]
        tokidx, label_ids = build_token_index(pm, labels, dev)
        cfg = get_cfg()
        cfg.MODEL.ATSS.DETECTIONS_PER_IMG = ndet
        cfg.TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM = mdetr
        B = gd["dot.0"].shape[0]
        T = gd["dot.0"].shape[2]
        head = {"dot": [], "bbox_reg": [], "centerness": [], "tbias": torch.zeros(B, T, device=dev)}
        for l in range(5):
            dot, reg, ctr = t(f"dot.{l}"), t(f"bbox_reg.{l}").to(H16), t(f"centerness.{l}").to(H16)
            head["dot"].append(dot.to(dev).contiguous())                        # fp32 logits straight into the kernel
            head["bbox_reg"].append(reg.to(dev))
            head["centerness"].append(ctr.to(dev))
            HW = dot.shape[1]
            r, cls = ops.align_scores(head["dot"][l], head["tbias"], tokidx, ctr.permute(0, 2, 3, 1).reshape(B, HW).contiguous().to(dev),
                                      cfg.MODEL.ATSS.INFERENCE_TH, want_cls=True)
            cls_ref = torch.stack([dot.sigmoid()[:, :, torch.tensor(pm[k])].mean(-1) for k in labels], -1)
            res.append(_stat(f"{name}: mq_align_scores class scores lvl{l}", cls, cls_ref, tol=1e-5))
            val_ref = torch.where(cls_ref > cfg.MODEL.ATSS.INFERENCE_TH, cls_ref * ctr.float().permute(0, 2, 3, 1).reshape(B, HW, 1).sigmoid(),
                                  torch.full_like(cls_ref, -1.0))
            # candidates exactly on the threshold may flip with the last ulp of the sigmoid: compare away from it
            far = (cls_ref - cfg.MODEL.ATSS.INFERENCE_TH).abs() > 1e-5
            res.append(_stat(f"{name}: mq_align_scores ranked values lvl{l}", r.cpu()[far], val_ref[far], tol=1e-5))
        anchors = [t(f"anchors.{l}").to(dev) for l in range(5)]
        post = pipeline.postprocess(cfg, head, anchors, sizes, tokidx, label_ids)
        # (a) against the reference's own outputs stored in the fixture: the product's head hands bbox_reg / centerness over in
        #     fp16, which moves a score by <= 2e-4 (sigmoid slope x 2^-11 x |ctr|) -- count and sorted scores are compared;
        # (b) against the oracle post-processor (itself pinned to the same fixture at 1e-5, tests/test_oracle_golden.py) fed
        #     with the SAME fp16-rounded bbox_reg / centerness: scores 1e-5, labels exact, boxes 1e-3 px.
        from dataclasses import replace
        from oracle import tiny_spec
        sp = replace(tiny_spec(), mdetr_class_num=mdetr, detections_per_img=ndet)
        odets = op.atss_postprocess([t(f"bbox_reg.{l}").to(H16).float() for l in range(5)], [t(f"centerness.{l}").to(H16).float() for l in range(5)],
                                    [t(f"dot.{l}") for l in range(5)], [t(f"anchors.{l}") for l in range(5)], sizes, pm, sp)
        for b in range(B):
            n = int(post["counts"][b])
            gb, gs, gl = post["boxes"][b, :n].cpu(), post["scores"][b, :n].cpu(), post["labels"][b, :n].cpu()
            o2 = torch.argsort(gs, descending=True, stable=True)
            for tag, rb, rs, rl, stol, btol in (("reference fixture", t(f"boxes{b}"), t(f"scores{b}"), t(f"labels{b}"), 3e-4, None),
                                                ("oracle on the same fp16 inputs", odets[b]["boxes"], odets[b]["scores"], odets[b]["labels"], 1e-5, 1e-3)):
                o1 = torch.argsort(rs, descending=True, stable=True)
                same_n = n == len(rb)
                res.append({"name": f"{name} vs {tag}: detection count img{b} ({n} vs {len(rb)})", "max_err": 0.0 if same_n else 1.0,
                            "mean_err": 0.0, "ref_absmax": 1.0, "norm_err": 0.0 if same_n else 1.0, "tol": 0.0, "ok": same_n})
                if not (same_n and n):
                    continue
                res.append(_stat(f"{name} vs {tag}: scores img{b}", gs[o2], rs[o1], tol=stol))
                if btol is not None:
                    lab_ok = bool(torch.equal(gl[o2], rl[o1]))
                    res.append({"name": f"{name} vs {tag}: labels img{b} (exact)", "max_err": 0.0 if lab_ok else 1.0, "mean_err": 0.0,
                                "ref_absmax": 1.0, "norm_err": 0.0 if lab_ok else 1.0, "tol": 0.0, "ok": lab_ok})
                    berr = (gb[o2] - rb[o1]).abs().max().item()
                    res.append({"name": f"{name} vs {tag}: boxes img{b} (px)", "max_err": berr, "mean_err": 0.0, "ref_absmax": 1.0,
                                "norm_err": berr, "tol": btol, "ok": berr <= btol})
    return res


def check_align_fused(dev):
    """mq_align_fused_fwd (box / centerness heads + dot-product alignment + sigmoid + token -> class aggregation + threshold + x centerness
    in one kernel, all levels) against a plain fp32 statement on the same 16-bit operands: ragged level sizes (tiles that straddle
    levels, a last tile with 3 rows), one caption per batch and one per item, every aggregation, live-token bounds that cut the
    text operand (kv_max) and T not a multiple of 16."""
    from mq_det_amd import ops
    res = []
    g = torch.Generator().manual_seed(77)
    cases = [dict(B=2, sizes=[(9, 13), (5, 7), (3, 4), (2, 2), (1, 2)], T=256, kv=141, L=40, MT=4, agg=0, per_item=False),
             dict(B=3, sizes=[(11, 12), (1, 3)], T=256, kv=0, L=7, MT=3, agg=1, per_item=True),
             dict(B=1, sizes=[(16, 8), (7, 5), (2, 1)], T=72, kv=50, L=5, MT=2, agg=2, per_item=False),
             dict(B=2, sizes=[(40, 50)], T=256, kv=17, L=3, MT=1, agg=0, per_item=False)][:3 if QUICK else 4]
    for c in cases:
        B, sizes, T, L, MT = c["B"], c["sizes"], c["T"], c["L"], c["MT"]
        N = sum(h * w for h, w in sizes)
        nv = c["kv"] if c["kv"] else T
        if ops.f32_operands() == 1 and nv > 144:
            continue        # precise mode on the device: 264 floats of LDS per live text token -- the pipeline sends longer captions down the GEMM path
        tok = (torch.randn(B, N, 256, generator=g) * 0.7).to(H16)
        tk = (torch.randn(B, T, 256, generator=g) * 0.12).to(H16)
        tbias = torch.randn(B, T, generator=g) * 0.5 - 1.0
        wbc = torch.zeros(16, 256)
        wbc[:5] = torch.randn(5, 256, generator=g) * 0.05
        wbc = wbc.to(H16)
        bbc = torch.cat([torch.randn(5, generator=g) * 0.1, torch.zeros(3)])
        scales = torch.rand(len(sizes), generator=g) + 0.5
        shape = (B, L, MT) if c["per_item"] else (L, MT)
        tokidx = torch.randint(0, nv, shape, generator=g).to(torch.int32)
        tokidx[..., 1:][torch.rand(tokidx[..., 1:].shape, generator=g) < 0.4] = -1                  # ragged label spans
        if L > 4:
            tokidx[..., 3, :] = -1                                                                  # a label without tokens: never scored
        out = ops.align_fused(tok.to(dev), tk.to(dev), tbias.to(dev), wbc.to(dev), bbc.to(dev), scales.to(dev), tokidx.to(dev), sizes, 0.05,
                              agg=c["agg"], kv_max=c["kv"], want_cls=True, want_logits=True)
        dots = torch.bmm(tok.float(), tk.float().transpose(1, 2))
        bc = tok.float() @ wbc.float().t()[:, :8] + bbc
        sig = (dots + tbias[:, None]).clamp(-50000, 50000).sigmoid()
        cls = torch.zeros(B, N, L)
        for b in range(B):
            tix = tokidx[b] if c["per_item"] else tokidx
            for l in range(L):
                toks = [int(t) for t in tix[l] if int(t) >= 0]
                if toks:
                    sel = sig[b][:, toks]
                    cls[b, :, l] = sel.mean(-1) if c["agg"] == 0 else (sel.max(-1)[0] if c["agg"] == 1 else sel.prod(-1) ** (1.0 / len(toks)))
        ranked = torch.where(cls > 0.05, (cls * bc[..., 4].sigmoid()[..., None]).clamp(min=1.17549435e-38), torch.full_like(cls, -1.0))
        tag = f"align_fused B={B} levels={len(sizes)} N={N} T={T} kv={c['kv']} L={L} agg={c['agg']} per_item={c['per_item']}"
        nvb = -(-nv // 16) * 16
        res.append(_stat(f"{tag}: logits (live text blocks)", out["logits"][:, :, :min(nvb, T)], dots[:, :, :min(nvb, T)], tol=1e-3))
        res.append(_stat(f"{tag}: centerness logits", out["ctr"], bc[..., 4], tol=1e-3))
        off = 0
        for l, (h, w) in enumerate(sizes):
            hw = h * w
            res.append(_stat(f"{tag}: level {l} box deltas", out["reg"][l], bc[:, off:off + hw, :4] * scales[l], tol=2e-3))
            res.append(_stat(f"{tag}: level {l} class scores", out["cls"][l], cls[:, off:off + hw], tol=1e-3))
            got, ref = out["ranked"][l].cpu(), ranked[:, off:off + hw]
            edge = (cls[:, off:off + hw] - 0.05).abs() < 1e-3                                         # a score ON the threshold may fall either way
            res.append(_stat(f"{tag}: level {l} ranked scores", torch.where(edge, ref, got), ref, tol=1e-3))
            off += hw
    return res


def check_score_agg(dev, golden_dir=None):
    """MODEL.DYHEAD.SCORE_AGG = MAX / ONEHOT / POWER (rpn/inference.py:772-824): mq_align_scores_fwd's class scores against the
    REFERENCE-generated fixture tests/golden/score_agg.npz (oracle/gen_golden_score_agg.py), and the whole product
    post-processing per mode against the oracle post-processor on the inputs of atss_post_*.npz."""
    import os
    import numpy as np
    from dataclasses import replace
    from oracle import postprocess as op, tiny_spec
    from mq_det_amd import ops, get_cfg
    from mq_det_amd.modeling import pipeline
    from mq_det_amd.modeling.query_selector import build_token_index
    golden_dir = golden_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

    def pmap_of(gd):
        pm, o = {}, 0
        for k, n in zip(gd["pmap_keys"].tolist(), gd["pmap_lens"].tolist()):
            pm[int(k)] = [int(v) for v in gd["pmap_flat"][o:o + n]]
            o += n
        return pm

    def index_for(pm, agg):
        if agg == "ONEHOT":
            n = len(pm)
            return build_token_index({j + 1: [j] for j in range(n)}, list(range(1, n + 1)), dev), list(range(1, n + 1))
        labels = [k for k, v in pm.items() if len(v)]
        return build_token_index(pm, labels, dev), labels

    res = []
    gd = np.load(os.path.join(golden_dir, "score_agg.npz"))
    pm = pmap_of(gd)
    logits, ctr = torch.from_numpy(gd["logits"]), torch.from_numpy(gd["centerness"])
    B, HW, T = logits.shape
    tb = torch.zeros(B, T, device=dev)
    for fam, aggs in (("dyhead", ("MEAN", "MAX", "ONEHOT")), ("mdetr", ("MEAN", "MAX", "ONEHOT", "POWER"))):
        for agg in aggs:
            (tokidx, _), labels = index_for(pm, agg)
            _, cls = ops.align_scores(logits.to(dev).contiguous(), tb, tokidx, ctr.to(H16).to(dev).contiguous(), 0.05, want_cls=True,
                                      agg=ops.SCORE_AGG[agg])
            ref = torch.from_numpy(gd[f"{fam}_{agg}"])[:, :, [k - 1 for k in labels]]
            res.append(_stat(f"score_agg {agg} ({fam}): mq_align_scores class scores vs reference fixture", cls, ref, tol=1e-5))
    for name, mdetr in (("atss_post_dyhead", -1), ("atss_post_mdetr", 3000)):
        gd = np.load(os.path.join(golden_dir, name + ".npz"))
        t = lambda k: torch.from_numpy(gd[k])                                   # noqa: E731
        pm = pmap_of(gd)
        sizes = [tuple(int(v) for v in r) for r in gd["sizes"]]
        B, T = gd["dot.0"].shape[0], gd["dot.0"].shape[2]
        for agg in ("MAX", "ONEHOT") + (("POWER",) if mdetr != -1 else ()):
            cfg = get_cfg()
            cfg.MODEL.ATSS.DETECTIONS_PER_IMG, cfg.TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM, cfg.MODEL.DYHEAD.SCORE_AGG = 100, mdetr, agg
            (tokidx, label_ids), _ = index_for(pm, agg)
            head = {"dot": [t(f"dot.{l}").to(dev).contiguous() for l in range(5)], "bbox_reg": [t(f"bbox_reg.{l}").to(H16).to(dev) for l in range(5)],
                    "centerness": [t(f"centerness.{l}").to(H16).to(dev) for l in range(5)], "tbias": torch.zeros(B, T, device=dev)}
            post = pipeline.postprocess(cfg, head, [t(f"anchors.{l}").to(dev) for l in range(5)], sizes, tokidx, label_ids)
            sp = replace(tiny_spec(), mdetr_class_num=mdetr, detections_per_img=100, score_agg=agg)
            odets = op.atss_postprocess([t(f"bbox_reg.{l}").to(H16).float() for l in range(5)], [t(f"centerness.{l}").to(H16).float() for l in range(5)],
                                        [t(f"dot.{l}") for l in range(5)], [t(f"anchors.{l}") for l in range(5)], sizes, pm, sp)
            for b in range(B):
                n = int(post["counts"][b])
                gb, gs, gl = post["boxes"][b, :n].cpu(), post["scores"][b, :n].cpu(), post["labels"][b, :n].cpu()
                rb, rs, rl = odets[b]["boxes"], odets[b]["scores"], odets[b]["labels"]
                o1, o2 = torch.argsort(rs, descending=True, stable=True), torch.argsort(gs, descending=True, stable=True)
                ok = n == len(rb) and n > 0 and bool(torch.equal(gl[o2], rl[o1]))
                res.append({"name": f"{name} SCORE_AGG={agg}: count + labels img{b} ({n} vs {len(rb)})", "max_err": 0.0 if ok else 1.0, "mean_err": 0.0,
                            "ref_absmax": 1.0, "norm_err": 0.0 if ok else 1.0, "tol": 0.0, "ok": ok})
                if ok:
                    res.append(_stat(f"{name} SCORE_AGG={agg}: scores img{b}", gs[o2], rs[o1], tol=1e-5))
                    res.append(_stat(f"{name} SCORE_AGG={agg}: boxes img{b}", gb[o2], rb[o1], tol=1e-5))
    return res


def check_layernorm(dev):
    """mq_layernorm_fwd in every precision combination: fp16 / fp32 input, fp16 / fp32 residual, fp32 second output."""
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(22)
    res = []
    for rows, C, eps in ((1000, 96, 1e-5), (777, 192, 1e-5), (130, 384, 1e-5), (65, 768, 1e-12), (50, 1536, 1e-5), (300, 256, 1e-5)):
        x = (torch.randn(rows, C, generator=g) * 2 + 0.5).to(H16)
        w, b = (torch.randn(C, generator=g) * 0.1 + 1).to(H16), (torch.randn(C, generator=g) * 0.1).to(H16)
        ref = F.layer_norm(x.float(), (C,), w.float(), b.float(), eps)
        res.append(_stat(f"layernorm rows={rows} C={C}", ops.layer_norm(x.to(dev), w.to(dev), b.to(dev), eps), ref))
        r = torch.randn(rows, C, generator=g).to(H16)                 # fp16 + fp16: LN(fp16(x + r)), sum returned in fp16
        xs = (x.float() + r.float()).to(H16)
        y, s_out = ops.layer_norm(x.to(dev), w.to(dev), b.to(dev), eps, residual=r.to(dev))
        res.append(_stat(f"add+layernorm rows={rows} C={C}: y", y, F.layer_norm(xs.float(), (C,), w.float(), b.float(), eps)))
        res.append(_stat(f"add+layernorm rows={rows} C={C}: fp16 sum (bit-exact)", s_out, xs.float(), tol=0.0))
        r32 = torch.randn(rows, C, generator=g) * 3                  # fp16 delta + fp32 stream: unrounded fp32 sum
        s32 = x.float() + r32
        y, y32, s_out = ops.layer_norm(x.to(dev), w.to(dev), b.to(dev), eps, residual=r32.to(dev), want_y32=True)
        ref32 = F.layer_norm(s32, (C,), w.float(), b.float(), eps)
        res.append(_stat(f"fp16 + fp32 stream rows={rows} C={C}: y", y, ref32))
        res.append(_stat(f"fp16 + fp32 stream rows={rows} C={C}: y32", y32, ref32, tol=2e-5))
        res.append(_stat(f"fp16 + fp32 stream rows={rows} C={C}: fp32 sum (bit-exact)", s_out, s32, tol=0.0))
        y32 = ops.layer_norm(r32.to(dev), w.to(dev), b.to(dev), eps, want_y=False, want_y32=True)      # fp32 in, fp32 out only
        res.append(_stat(f"fp32 in rows={rows} C={C}: y32", y32, F.layer_norm(r32, (C,), w.float(), b.float(), eps), tol=2e-5))
    return res


def check_conv3x3(dev):
    """Implicit-GEMM 3x3 conv vs F.conv2d (fp32, CPU): strides, N = 256 / 27, sizes that straddle tiles and images."""
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(21)
    res = []
    for (B, H, W, N, stride) in ((2, 13, 21, 256, 1), (3, 25, 42, 256, 2), (2, 100, 168, 27, 1), (1, 7, 11, 27, 1), (2, 50, 84, 256, 1)):
        if QUICK and H * W > 2000:
            continue
        x = torch.randn(B, 256, H, W, generator=g).to(H16)
        w = (torch.randn(N, 256, 3, 3, generator=g) / 48).to(H16)
        bias = torch.randn(N, generator=g).to(H16)
        ref = F.conv2d(x.float(), w.float(), bias.float(), stride=stride, padding=1)
        wp = w.permute(0, 2, 3, 1).reshape(N, -1)
        rows = 32 if N <= 32 else 256
        wp = torch.cat([wp, wp.new_zeros(rows - N, wp.shape[1])], 0).contiguous()
        y = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().to(dev), wp.to(dev), bias.to(dev), N, stride)
        res.append(_stat(f"conv3x3 igemm B={B} {H}x{W} N={N} s={stride}", y.permute(0, 3, 1, 2), ref, tol=3e-3))
        if N <= 32 and stride == 1:       # LDS-window kernel of the offset conv: fp32 NCHW out; also from a strided level view
            y2 = ops.conv3x3_nchw32(x.permute(0, 2, 3, 1).contiguous().to(dev), wp.to(dev), bias.to(dev), N)
            res.append(_stat(f"conv3x3 LDS-window (fp32 NCHW) B={B} {H}x{W} N={N}", y2, ref, tol=2e-3))
            big = torch.zeros(B, H * W + 37, 256, dtype=H16, device=dev)
            big[:, 5:5 + H * W] = x.permute(0, 2, 3, 1).reshape(B, H * W, 256).to(dev)
            y3 = ops.conv3x3_nchw32(big[:, 5:5 + H * W].reshape(B, H, W, 256), wp.to(dev), bias.to(dev), N)
            res.append(_stat(f"conv3x3 LDS-window, level view of a token buffer B={B} {H}x{W}", y3, ref, tol=2e-3))
    res += check_conv3x3_group(dev)
    return res


def check_dyconv_epilogue_group(dev):
    """mq_dyconv_epilogue_group (csrc/dyconv.hip): a DyConv layer of the tiny model with its epilogue (GroupNorm affine + up-sampling +
    scale attention + branch mean, DYReLU coefficients) as two launches for all levels == the same layer through the per-level launches,
    bit for bit (same bodies, same summation order), with the DYReLU deferred and applied; pyramids with 1 .. 3 branches per level."""
    from mq_det_amd import ops
    from mq_det_amd.modeling import pipeline
    spec, sd, cfg, model, P = tiny(dev)
    g = torch.Generator().manual_seed(33)
    res = []
    b = "rpn.head.dyhead_tower.2"
    saved = ops.KERNELS["DYCONV_EPILOGUE_GROUPED"]
    try:
        pyramids = ([(20, 24), (10, 12), (5, 6), (3, 3), (2, 2)], [(17, 9), (9, 5)], [(13, 21)])
        for pi, sizes in enumerate(pyramids[:1 if (QUICK and H16 != torch.float16) else (2 if QUICK else 3)]):
            feats = [torch.randn(2, 256, h, w, generator=g).to(H16).to(dev).contiguous(memory_format=torch.channels_last) for h, w in sizes]
            tok, szs = pipeline._to_tokens(feats)
            outs = {}
            with torch.no_grad():
                for mode in (0, 1):
                    ops.KERNELS["DYCONV_EPILOGUE_GROUPED"] = mode
                    outs[mode] = (pipeline.dyconv_tokens(P, cfg, b, tok.contiguous(), szs, defer_relu=True),
                                  pipeline.dyconv_tokens(P, cfg, b, tok.contiguous(), szs) if (pi == 0 or not QUICK) else None)
            (pre0, c0), o0 = outs[0]
            (pre1, c1), o1 = outs[1]
            res.append(_stat(f"dyconv epilogue grouped == per level, {len(sizes)} levels: output before DYReLU (exact)", pre1, pre0.float().cpu(), tol=0.0))
            res.append(_stat(f"dyconv epilogue grouped == per level, {len(sizes)} levels: DYReLU coefficients (exact)", c1, c0.float().cpu(), tol=0.0))
            if o0 is not None:
                res.append(_stat(f"dyconv epilogue grouped == per level, {len(sizes)} levels: output with DYReLU applied (exact)", o1, o0.float().cpu(), tol=0.0))
    finally:
        ops.KERNELS["DYCONV_EPILOGUE_GROUPED"] = saved
    return res


def check_conv3x3_group(dev):
    """mq_conv3x3_nchw32_group_fwd (csrc/conv_small3.hip): the 27-channel offset conv of every level of a pyramid in one launch (levels =
    slices of ONE token buffer, as dyconv_tokens passes them) vs F.conv2d per level, and vs the per-level kernel (same products, other
    fp32 summation order).  Pyramids: the tiny model's, one with tile-edge sizes (8 x 16 tiles: 8 / 9 rows, 16 / 17 columns), a single
    level, and B = 3 so that the workgroups' tile runs cross image and level boundaries."""
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(31)
    res = []
    pyramids = [(2, [(20, 24), (10, 12), (5, 6), (3, 3), (2, 2)]), (3, [(9, 17), (8, 16), (1, 1)]), (1, [(7, 11)])]
    if not QUICK:
        pyramids.append((2, [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]))
    for B, sizes in pyramids:
        n = sum(h * w for h, w in sizes)
        tok = torch.randn(B, n, 256, generator=g).to(H16)
        w = (torch.randn(27, 256, 3, 3, generator=g) / 48).to(H16)
        bias = torch.randn(27, generator=g).to(H16)
        wp = w.permute(0, 2, 3, 1).reshape(27, -1)
        wp = torch.cat([wp, wp.new_zeros(5, wp.shape[1])], 0).contiguous()
        td, off, lv = tok.to(dev), 0, []
        for (h, ww) in sizes:
            lv.append(td[:, off:off + h * ww].reshape(B, h, ww, 256))
            off += h * ww
        if ops.f32_operands() == 1:
            break                 # precise mode on the device: the grouped kernel's window does not fit at fp32; the per-level kernel above is what runs
        assert ops.conv3x3_nchw32_group_supported(lv, 27)
        got = ops.conv3x3_nchw32_group(lv, wp.to(dev), bias.to(dev), 27)
        for l, (x, y) in enumerate(zip(lv, got)):
            ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float(), bias.float(), padding=1)
            res.append(_stat(f"conv3x3 group B={B} level {l} {sizes[l][0]}x{sizes[l][1]}: vs F.conv2d", y, ref, tol=2e-3))
            one = ops.conv3x3_nchw32(x, wp.to(dev), bias.to(dev), 27)
            res.append(_stat(f"conv3x3 group B={B} level {l} {sizes[l][0]}x{sizes[l][1]}: vs the per-level kernel (fp32 summation order)", y, one.float().cpu(), tol=2e-6))
    return res


def check_dyconv(dev):
    from oracle import head as oh
    from mq_det_amd.modeling import pipeline
    spec, sd, cfg, model, P = tiny(dev)
    g = torch.Generator().manual_seed(16)
    sizes = [(20, 24), (10, 12), (5, 6), (3, 3), (2, 2)]
    feats = [torch.randn(2, 256, h, w, generator=g).to(H16) for h, w in sizes]
    b = "rpn.head.dyhead_tower.2"
    with torch.no_grad():
        ref = oh.dyconv(sd, b, [f.float() for f in feats], spec)
        x = [f.to(dev).contiguous(memory_format=torch.channels_last) for f in feats]
        got = pipeline.dyconv(P, cfg, b, x)
        res = [_stat(f"dyconv (grouped fused DCNv2) lvl{i}", got[i], ref[i], tol=2.5e-3) for i in range(5)]
        # the same layer with its DYReLU left to the next layer's LayerNorm (mq_dyrelu_ln_fwd): LN(dyconv(x)) of the oracle
        from mq_det_amd import ops
        tok, szs = pipeline._to_tokens(x)
        pre, coef = pipeline.dyconv_tokens(P, cfg, b, tok.contiguous(), szs, defer_relu=True)
        gw, gb = P["rpn.head.dyhead_tower.3.b_attn.layer_norm_v.weight"], P["rpn.head.dyhead_tower.3.b_attn.layer_norm_v.bias"]
        got_ln = ops.dyrelu_layer_norm(pre, coef, szs, gw, gb, 1e-5)
        ref_tok = torch.cat([r.flatten(2).transpose(1, 2) for r in ref], 1)
        ref_ln = F.layer_norm(ref_tok, (256,), gw.float().cpu(), gb.float().cpu(), 1e-5)
        res.append(_stat("dyconv with DYReLU applied by the next LayerNorm: LN(dyconv(x)), all levels", got_ln, ref_ln, tol=2.5e-3))
        unf = ops.layer_norm(torch.cat([g_.flatten(2).transpose(1, 2) for g_ in got], 1).contiguous(), gw, gb, 1e-5)
        res.append(_stat("... vs the stand-alone DYReLU pass + LayerNorm of the product", got_ln, unf.float().cpu(), tol=1e-2))
    return res


def check_nms(dev):
    from oracle import postprocess as op
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(17)
    res = []
    for n in (0, 1, 63, 64, 65, 700, 4100):
        B = 2
        N = max(n, 1)
        xy = torch.rand(B, N, 2, generator=g) * 300
        wh = torch.rand(B, N, 2, generator=g) * 120 + 4
        boxes = torch.cat([xy, xy + wh], -1)
        scores = torch.rand(B, N, generator=g)
        labels = torch.randint(1, 6, (B, N), generator=g).int()
        order = torch.argsort(scores, dim=1, descending=True, stable=True)
        boxes = torch.gather(boxes, 1, order[:, :, None].expand(-1, -1, 4)).contiguous()
        labels = torch.gather(labels, 1, order).contiguous()
        scores = torch.gather(scores, 1, order)
        nvalid = torch.tensor([n, max(n - 7, 0)], dtype=torch.int32)
        keep = ops.ml_nms(boxes.to(dev), labels.to(dev), nvalid.to(dev), 0.6).cpu()
        ok = True
        for b in range(B):
            nv = int(nvalid[b])
            ref = torch.zeros(N, dtype=torch.bool)
            if nv:
                ref[op.ml_nms(boxes[b, :nv], scores[b, :nv], labels[b, :nv].float(), 0.6)] = True
            ok &= bool(torch.equal(ref, keep[b]))
        res.append({"name": f"ml_nms n={n}", "max_err": 0.0 if ok else 1.0, "mean_err": 0.0, "ref_absmax": 1.0,
                    "norm_err": 0.0 if ok else 1.0, "tol": 0.0, "ok": ok})
    return res


def make_inputs(spec, B=2, hw=((150, 190), (160, 170)), nvalid=30, seed=3):
    from oracle import detector as od
    from oracle.weights import make_query_bank
    g = torch.Generator().manual_seed(seed)
    imgs = [torch.randn(3, h, w, generator=g).to(H16).float() for (h, w) in hw[:B]]
    images, sizes = od.pad_images(imgs, spec.size_divisibility)
    T = spec.max_query_len
    ids = torch.zeros(B, T, dtype=torch.long)
    ids[:, :nvalid] = torch.randint(1, spec.vocab, (nvalid,), generator=g)[None]
    am = torch.zeros(B, T, dtype=torch.long)
    am[:, :nvalid] = 1
    pm = {1: [1], 2: [3, 4], 3: [6], 4: [8, 9, 10], 5: [12], 6: [14, 15]}
    bank = make_query_bank(pm.keys(), spec)
    return images, sizes, ids, am, pm, bank


def check_full_model(dev, vision_queries=True, large=False):
    """Whole forward, tiny-depth MQ-GLIP (real widths / head dims), fp16 HIP path vs fp32 oracle.
    vision_queries=False: plain GLIP path (no query bank -> no pre-select / GCP blocks), BASELINE configs[0] shape.
    large: Swin-L backbone (window 12) -- the MQ-GLIP-L model family at tiny depth (BASELINE configs[3])."""
    from oracle import detector as od
    from mq_det_amd.structures import ImageList
    spec, sd, cfg, model, P = tiny(dev, large)
    images, sizes, ids, am, pm, bank = make_inputs(spec)
    if not vision_queries:
        bank = None
        images, sizes = images[:1], sizes[:1]
        ids, am = ids[:1], am[:1]
    model.load_query_bank(bank)
    with torch.no_grad():
        dets, inter = od.forward(sd, spec, images, sizes, ids, am, pm, bank, return_intermediates=True)
        raw = model(ImageList(images.to(dev), sizes), captions=None, positive_map=pm, return_raw=True,
                    input_ids=ids.to(dev), attention_mask=am.to(dev))
    res = [_stat(f"full: fpn p{i + 3}", raw["feats"][i], inter["fpn"][i], tol=6e-3) for i in range(5)]
    # padding-token rows are dead (never keys, never scored; since round 5 the device programs do not compute them at all: live-row
    # compaction, detector._live_len): compare live rows
    live = am.bool()
    res.append(_stat("full: language hidden", raw["lang"]["hidden"].cpu()[live], inter["lang"]["hidden"][live], tol=2e-2))
    h = inter["head"]
    res.append(_stat("full: head text hidden (caption tokens)", raw["head"]["hidden"].cpu()[live], h["hidden"][live], tol=1.5e-2))
    nv = int(am[0].sum())
    for l in range(5):
        res.append(_stat(f"full: head feats lvl{l}", raw["head"]["feats"][l], h["feats"][l], tol=2.6e-2))
        res.append(_stat(f"full: bbox_reg lvl{l}", raw["head"]["bbox_reg"][l], h["bbox_reg"][l], tol=2e-2))
        res.append(_stat(f"full: centerness lvl{l}", raw["head"]["centerness"][l], h["centerness"][l], tol=1.5e-2))
        logit = raw["head"]["dot"][l].float() + raw["head"]["tbias"][:, None, :]
        res.append(_stat(f"full: dot logits lvl{l}", logit[:, :, :nv], h["dot_product_logits"][l][:, :, :nv], tol=1.8e-2))
        cls_ref = torch.stack([h["dot_product_logits"][l].sigmoid()[:, :, torch.tensor(pm[k])].mean(-1) for k in pm], -1)
        # sigmoid scores in [0, 1]: logits drift ~0.1-0.2 (fp16 through the whole stack) x slope 0.25
        res.append(_stat(f"full: class scores lvl{l}", raw["post"]["cls"][l], cls_ref, tol=3.3e-2))
    # final detections: match by IoU against the oracle's detections
    post = raw["post"]
    for b in range(len(dets)):
        n = int(post["counts"][b])
        gb, gs, gl = post["boxes"][b, :n].cpu(), post["scores"][b, :n].cpu(), post["labels"][b, :n].cpu()
        rb, rs, rl = dets[b]["boxes"], dets[b]["scores"], dets[b]["labels"]
        top = torch.argsort(rs, descending=True)[:50]
        matched = 0
        for i in top.tolist():
            same = (gl == rl[i])
            if not same.any():
                continue
            lt = torch.max(gb[:, :2], rb[i, :2])
            br = torch.min(gb[:, 2:], rb[i, 2:])
            inter_a = (br - lt + 1).clamp(min=0).prod(1)
            a1 = (gb[:, 2] - gb[:, 0] + 1) * (gb[:, 3] - gb[:, 1] + 1)
            a2 = (rb[i, 2] - rb[i, 0] + 1) * (rb[i, 3] - rb[i, 1] + 1)
            iou = torch.where(same, inter_a / (a1 + a2 - inter_a), torch.zeros_like(a1))
            j = int(iou.argmax())
            # bf16: scores drift 8x further (3 fewer significant bits per operand), boxes move with them
            if iou[j] > (0.9 if TOL_SCALE == 1.0 else 0.8) and abs(float(gs[j] - rs[i])) < 0.02 * TOL_SCALE:
                matched += 1
        frac = matched / max(1, len(top))
        res.append({"name": f"full: top-50 detections matched (IoU>{0.9 if TOL_SCALE == 1.0 else 0.8}, |ds|<{0.02 * TOL_SCALE:g}) img{b} n_hip={n} n_ref={len(rb)}",
                    "max_err": 1 - frac, "mean_err": 0.0, "ref_absmax": 1.0, "norm_err": 1 - frac, "tol": 0.2, "ok": frac >= 0.8})
    return res


def all_checks(dev):
    """Every parity check, flattened (name -> result)."""
    out = []
    specs = [
        dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True),
        dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True, clamp=50000.0, big=True),
        dict(B=1, H=8, D=32, Nq=200, Nk=5577, nsplit=4),
        dict(B=1, H=8, D=32, Nq=37, Nk=61),
        dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True, kvlen=True),
    ]
    for s in specs:
        out.append(("attention", lambda s=s: check_attention(dev, **s)))
    out += [("attention", lambda: check_attention_strided(dev)),
            ("attention", lambda: check_attention_text(dev)),
            ("attention", lambda: check_bert_attn_qkv(dev)),
            ("gcp", lambda: check_gcp_attn_fused(dev)),
            ("swin", lambda: check_patch_embed(dev)),
            ("post", lambda: check_post_fused(dev)),
            ("dyconv", lambda: check_dyconv_epilogue_group(dev)),
            ("window_attn", lambda: check_window_attention(dev)),
            ("swin_fpn", lambda: check_swin_fpn(dev)),
            ("gcp", lambda: check_gcp_block(dev)),
            ("gcp", lambda: check_pre_select(dev)),
            ("bert", lambda: check_bert_layer(dev, False)),
            ("bert", lambda: check_bert_layer(dev, True)),
            ("bert", lambda: check_bert_clamp_fused(dev)),
            ("vlfuse", lambda: check_vlfuse_kernels(dev)),
            ("vlfuse", lambda: check_vl_fuse(dev)),
            ("dcn", lambda: check_dcn(dev)),
            ("ref-pin", lambda: check_ref_pins(dev)),
            ("post", lambda: check_post_golden(dev)),
            ("post", lambda: check_score_agg(dev)),
            ("post", lambda: check_align_fused(dev)),
            ("swin", lambda: check_swin_mlp(dev)),
            ("gdino", lambda: check_msdeform_attn(dev)),
            ("roi", lambda: check_roi_align(dev)),
            ("roi", lambda: check_extract_query(dev)),
            ("conv", lambda: check_conv3x3(dev)),
            ("layernorm", lambda: check_layernorm(dev)),
            ("dyconv", lambda: check_dyconv(dev)),
            ("nms", lambda: check_nms(dev)),
            ("swin-L", lambda: check_window_attention(dev, large=True)),
            ("swin-L", lambda: check_swin_fpn(dev, large=True)),
            ("full", lambda: check_full_model(dev)),
            ("full-L", lambda: [dict(r, name=r["name"].replace("full:", "MQ-GLIP-L (tiny depth):")) for r in check_full_model(dev, large=True)]),
            ("full-novq", lambda: [dict(r, name=r["name"].replace("full:", "GLIP (no vision queries) B=1:")) for r in check_full_model(dev, False)])]
    import gdino_checks as gc
    out += [("gdino", lambda: gc.check_attention_qk_mask(dev)), ("gdino", lambda: gc.check_vlfuse_heads_mask(dev)),
            ("gdino", lambda: gc.check_msdeform_attn_q(dev)), ("gdino", lambda: gc.check_gdino_tiny(dev)),
            ("gdino", lambda: gc.check_gdino_state_dict_and_quirks(dev))]
    return out


# ----------------------------------------------------------------------------------------------------------------------
# Full-depth MQ-GLIP-T at the BENCHMARK configuration (bench.py / BASELINE.json configs[1]): Swin 2-2-6-2, 12 BERT layers
# with 6 GCP blocks, 6 fusion layers, 800x1333 images -- HIP path vs the fp32 oracle, with a per-stage error ladder.
#
# Stated tolerance.  The north-star asks for 1e-3; per KERNEL (same fp16-rounded operands) every check above meets it.
# End to end it cannot be met by ANY fp16-operand MFMA implementation of this (randomly initialised) network: rounding only
# the GEMM operands to fp16 in the fp32 oracle and nothing else (oracle/precision.py, the "fp16-operand floor") already moves
# the alignment logits by 1e-2 ... 3e-2 of their range (mean |error| 0.03 ... 0.09 on |logit| <= 18; the network amplifies a
# relative perturbation ~300x: fp32 and fp64 oracles differ by 1e-5).  Measured on MI355X (profiles/r02_error_ladder.txt):
# with fp32 residual streams the product sits at 1.2 - 1.4x that floor at EVERY stage (mean error), i.e. 10 - 35x above
# 1e-3 at the heads and 2 - 3x above it after the backbone; what remains above the floor is fp16 storage of GEMM outputs.
# The tolerances below are the measured normalised errors (max |err| / max(1, max |ref|)) plus a ~2x margin; every per-kernel / tiny-model
# tolerance in this file was reset in round 4 to max(1e-3, ~2 x the device measurement of round 3), and the measured-gate of _stat()
# applies the same rule row by row from tests/golden/device_measured.json.
# Class scores live in [0, 1]: their MAX error is one worst location (floor: 0.09 on P7), the mean error is 3e-4 ... 1e-2.
BENCH_TOL = {"swin": 5e-3, "fpn": 5e-3, "pooled": 5e-3, "lang": 3e-2, "text": 3e-2, "feat": 5e-2, "box": 6e-2, "dot": 7e-2,
             "cls": 0.2}
# THE GATE.  tests/golden/floor_bench.json holds, per benchmark case and stage, the error of the fp16- (bf16-)
# OPERAND FLOOR against the fp32 oracle (oracle/gen_golden_floor.py: the oracle with only its contraction operands rounded,
# oracle/precision.py -- the smallest error any MFMA path with 16-bit operands can have on these weights).  Per stage row
#       ratio_mean = mean|hip - ref| / mean|floor - ref|          ratio_max = max|hip - ref| / max|floor - ref|
# and a case passes when
#   (a) the MEDIAN of ratio_mean over its ~50 stage rows is <= FLOOR_RATIO_MEDIAN (the product adds at most ~half the floor's own
#       error on top of it, taken over the whole depth of the model: a kernel that loses precision moves every row downstream
#       of it, and with them the median), and
#   (b) EVERY row has ratio_mean <= FLOOR_RATIO_ROW and ratio_max <= FLOOR_RATIO_ROW_MAX (no single stage is off by more than the scatter
#       below; the max is ONE worst element out of 10^5 .. 10^7 and moves by 2x between two equally good roundings: GPU call 5 saw 3.6
#       on "language hidden" with every other row of the case below 2.1).
# Why not a tight bound per row: the ratio of ONE row is a chaotic statistic.  profiles/r03_call4_bisect.txt runs the same case
# under ten kernel selections that are each exact to 1e-5 with fp32 operands (tests/test_simt_fp32_operands_cpu.py): with
# the backbone rows identical (swin c5 / fpn p3 at 1.32 in all ten) "dot-product logits lvl0" lands anywhere in 1.63 ... 2.31 and
# "language hidden" in 1.36 ... 1.66 -- any change of rounding ORDER upstream re-draws the row, the network amplifies a relative
# perturbation ~300x (fp32 vs fp64 oracles: 1e-5).  The median over the rows moves less: 1.23 / 1.41 / 1.24 for the three fp16
# cases and 1.56 for bf16 with the default kernels (profiles/r03_call3_ladder_summary.txt), 1.35 ... 1.65 over the ten selections of
# the bisect.  That spread is the resolution of ANY end-to-end statistic on this network: what the gate catches is a lost
# digit (a 16-bit stream where an fp32 one belongs, a wrong rounding mode, a mis-scaled operand); a kernel that is a few ulp
# worse is caught where it can be -- per kernel at 1e-3 on identical operands (every check above) and with fp32 operands at
# 1e-5 through the whole model (tests/test_simt_fp32_operands_cpu.py).  bf16 carries 8 mantissa bits in every STORED tensor as
# well (the floor only rounds operands), its factors are wider: _BF16_GATE.
# BENCH_TOL above stays as an absolute backstop (product error <= measured x 2, as in round 2).
FLOOR_RATIO_MEDIAN, FLOOR_RATIO_ROW, FLOOR_RATIO_ROW_MAX = 1.55, 3.0, 5.0      # median: measured 1.23 / 1.41 / 1.24 (round 3: 1.75)
_BF16_GATE = (1.9, 4.5, 6.0)            # median measured 1.55
_LADDER = {}
_FLOOR = None


def bench_case_key(family, caption, hw, dtype=None):
    dtype = dtype or H16
    return f"{family}|{caption}|" + "+".join(f"{h}x{w}" for h, w in hw) + "|" + str(dtype).replace("torch.", "")


def floor_fixture():
    global _FLOOR
    if _FLOOR is None:
        import json
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "floor_bench.json")
        _FLOOR = json.load(open(path)) if os.path.exists(path) else {}
    return _FLOOR


def caption_ids(spec, B, n_classes=40, words=(1, 2, 3, 4, 3, 2), seed=5):
    """Token ids / mask / positive_map of a synthetic class-name caption ([CLS] name . name . ... [SEP]) without a tokenizer."""
    g = torch.Generator().manual_seed(seed)
    T = spec.max_query_len
    pos, pm = 1, {}
    for i in range(n_classes):
        n = words[i % len(words)]
        pm[i + 1] = list(range(pos, pos + n))
        pos += n + 1                                   # + separator (the last one is replaced by [SEP])
    nvalid = pos
    ids = torch.zeros(B, T, dtype=torch.long)
    ids[:, :nvalid] = torch.randint(1000, spec.vocab, (nvalid,), generator=g)[None]
    am = torch.zeros(B, T, dtype=torch.long)
    am[:, :nvalid] = 1
    return ids, am, pm, nvalid


def bench_spec(family):
    from oracle import glip_l_spec, glip_t_spec
    return glip_l_spec() if family == "l" else glip_t_spec()


def bench_inputs(spec, caption, hw):
    """Seeded inputs of one benchmark case (shared by the product check and oracle/gen_golden_floor.py)."""
    from oracle import detector as od
    from oracle.weights import make_query_bank
    B = len(hw)
    words = {"short": (1,), "long": (1, 2, 3, 4, 3, 2), "xlong": (4,)}[caption]
    ids, am, pm, nv = caption_ids(spec, B, 40, words)
    g = torch.Generator().manual_seed(7)
    imgs = [torch.randn(3, h, w, generator=g).to(H16).float() for (h, w) in hw]
    images, sizes = od.pad_images(imgs, spec.size_divisibility)
    bank = make_query_bank(pm.keys(), spec)
    return images, sizes, ids, am, pm, nv, bank


def oracle_rows(inter, am, pm, nv):
    """Ordered {stage name: (kind, tensor)} of an oracle forward's intermediates -- the stages the ladder compares."""
    rows = {}
    live = am.bool()
    for i in range(3):
        rows[f"swin c{i + 3}"] = ("swin", inter["swin"][i + 1])
    for i in range(5):
        rows[f"fpn p{i + 3}"] = ("fpn", inter["fpn"][i])
    if inter.get("pooled") is not None:
        rows["pooled fpn tokens"] = ("pooled", inter["pooled"])
    rows["language hidden (caption tokens)"] = ("lang", inter["lang"]["hidden"][live])
    for i, o in enumerate(inter["head_trace"]):
        rows[f"head layer {i}: image tokens after VLFuse"] = ("feat", torch.cat([f.flatten(2).transpose(1, 2) for f in o["fuse_feats"]], 1))
        rows[f"head layer {i}: text hidden after BERT layer"] = ("text", o["bert_hidden"][live])
        rows[f"head layer {i}: image tokens after DyConv"] = ("feat", torch.cat([f.flatten(2).transpose(1, 2) for f in o["dyconv_feats"]], 1))
    h = inter["head"]
    for l in range(5):
        rows[f"bbox_reg lvl{l}"] = ("box", h["bbox_reg"][l])
        rows[f"centerness lvl{l}"] = ("box", h["centerness"][l])
        rows[f"dot-product logits lvl{l}"] = ("dot", h["dot_product_logits"][l][:, :, :nv])
        rows[f"class scores lvl{l}"] = ("cls", torch.stack([h["dot_product_logits"][l].sigmoid()[:, :, torch.tensor(pm[k])].mean(-1) for k in pm], -1))
    return rows


def _bench_model(dev, residual_fp32=True, family="t"):
    key = ("bench", residual_fp32, family)
    if key not in _CACHE:
        from oracle.weights import make_state_dict
        from mq_det_amd import get_cfg
        from mq_det_amd.modeling.detector import GeneralizedVLRCNN_New
        spec = bench_spec(family)
        sd = _CACHE.get(("bench_sd", family))
        if sd is None:
            sd = _CACHE[("bench_sd", family)] = make_state_dict(spec, 0)
        cfg = get_cfg()
        cfg.MODEL.SWINT.EMBED_DIM, cfg.MODEL.SWINT.NUM_HEADS = spec.swin_embed, spec.swin_heads
        cfg.MODEL.SWINT.WINDOW_SIZE, cfg.MODEL.SWINT.OUT_CHANNELS = spec.window, spec.swin_dims
        cfg.MODEL.SWINT.DEPTHS = spec.swin_depths
        cfg.MODEL.DYHEAD.NUM_CONVS = spec.dyhead_convs
        cfg.MODEL.DYHEAD.NUM_CLASSES = spec.num_classes
        cfg.MODEL.ATSS.DETECTIONS_PER_IMG = spec.detections_per_img
        cfg.TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM = spec.mdetr_class_num
        cfg.MODEL.RESIDUAL_FP32 = residual_fp32
        cfg.MODEL.COMPUTE_DTYPE = _DTYPE_NAME[H16]
        model = GeneralizedVLRCNN_New(cfg, tokenizer=object())
        model.load_state_dict(sd, strict=True)
        model.to(dev)
        model.prepare(dev)
        _CACHE[key] = (spec, sd, cfg, model)
    return _CACHE[key]


def _match_detections(gb, gs, gl, rb, rs, rl, top=100, iou_thr=0.9, ds=0.03):
    """Fraction of the oracle's `top` best detections that the HIP path reproduces (same label, IoU > iou_thr, |ds| < ds)."""
    order = torch.argsort(rs, descending=True)[:top]
    hit = 0
    for i in order.tolist():
        same = gl == rl[i]
        if not same.any():
            continue
        lt, br = torch.max(gb[:, :2], rb[i, :2]), torch.min(gb[:, 2:], rb[i, 2:])
        inter = (br - lt + 1).clamp(min=0).prod(1)
        a1 = (gb[:, 2] - gb[:, 0] + 1) * (gb[:, 3] - gb[:, 1] + 1)
        a2 = (rb[i, 2] - rb[i, 0] + 1) * (rb[i, 3] - rb[i, 1] + 1)
        iou = torch.where(same, inter / (a1 + a2 - inter), torch.zeros_like(a1))
        ok = (iou > iou_thr) & ((gs - rs[i]).abs() < ds)
        hit += bool(ok.any())
    return hit / max(1, len(order))


def _bench_got(raw, cg, am, nv, item=None):
    """The product's tensors (a return_raw forward + the Swin stages) under the oracle's stage names; item: keep one batch item."""
    live = am.bool()
    sl = (lambda t: t) if item is None else (lambda t: t[item:item + 1])
    got = {}
    for i in range(3):
        got[f"swin c{i + 3}"] = sl(cg[i]).permute(0, 3, 1, 2)
    for i in range(5):
        got[f"fpn p{i + 3}"] = sl(raw["feats"][i])
    got["pooled fpn tokens"] = sl(raw["pooled"])
    lh = raw["lang"]["hidden32"] if raw["lang"].get("hidden32") is not None else raw["lang"]["hidden"]
    got["language hidden (caption tokens)"] = sl(lh).cpu()[sl(live)]
    for i, a in enumerate(raw["head_trace"]):
        got[f"head layer {i}: image tokens after VLFuse"] = sl(a["fuse_tok"])
        got[f"head layer {i}: text hidden after BERT layer"] = sl(a["bert_hidden"]).cpu()[sl(live)]
        got[f"head layer {i}: image tokens after DyConv"] = sl(a["dyconv_tok"])
    for l in range(5):
        got[f"bbox_reg lvl{l}"] = sl(raw["head"]["bbox_reg"][l])
        got[f"centerness lvl{l}"] = sl(raw["head"]["centerness"][l])
        got[f"dot-product logits lvl{l}"] = (sl(raw["head"]["dot"][l]).float() + sl(raw["head"]["tbias"])[:, None, :])[:, :, :nv]
        got[f"class scores lvl{l}"] = sl(raw["post"]["cls"][l])
    return got


def _bench_gate(tag, got, ref_rows, fixture, fl_rows, residual_fp32, case):
    """Stage rows vs the oracle under the floor gate (FLOOR_RATIO_*): per-row records + the median record."""
    res, ratios = [], []
    med_gate, row_gate, row_max_gate = (FLOOR_RATIO_MEDIAN, FLOOR_RATIO_ROW, FLOOR_RATIO_ROW_MAX) if H16 == torch.float16 else _BF16_GATE
    for name, (kind, ref) in ref_rows.items():
        r = _stat(f"{tag} {name}", got[name], ref, tol=BENCH_TOL[kind])
        if H16 == torch.float32:
            # the precise mode: no operand rounding, so no floor to relate to -- every stage at F32_TOL = 1e-3 of the range AND zero elements
            # outside atol = rtol = 1e-3 (both inside _stat)
            r["gate"] = "fp32 operands: norm_err <= 1e-3, no element outside 1e-3 + 1e-3 |ref|"
            res.append(r)
            continue
        fx = fixture.get(name)
        if fl_rows is not None:
            f = _stat("floor", fl_rows[name][1], ref)
            fx = {"max": f["max_err"], "mean": f["mean_err"], "norm": f["norm_err"], "n": ref.numel()}
        if fx is not None:
            r["floor_norm_err"], r["floor_mean_err"] = fx["norm"], fx["mean"]
            r["ratio_mean"] = r["mean_err"] / max(fx["mean"], 1e-12)
            r["ratio_max"] = r["max_err"] / max(fx["max"], 1e-12)
            r["gate"] = f"mean <= {row_gate} x floor, max <= {row_max_gate} x floor"
            r["ok"] = bool(r["ok"] and r["ratio_mean"] <= row_gate and r["ratio_max"] <= row_max_gate)
            ratios.append(r["ratio_mean"])
        elif residual_fp32:
            r["ok"], r["gate"] = False, f"no floor fixture for case {case!r} / stage {name!r}: run python -m oracle.gen_golden_floor"
        res.append(r)
    if ratios:
        med = float(torch.tensor(ratios).median())
        res.append({"name": f"{tag} MEDIAN over {len(ratios)} stages of mean|hip - ref| / mean|floor - ref|", "max_err": med, "mean_err": med,
                    "ref_absmax": 1.0, "norm_err": med, "tol": med_gate, "ok": med <= med_gate, "ratio_mean": med, "ratio_max": med,
                    "gate": f"median <= {med_gate}"})
    return res


def check_benchmark_config(dev, caption="long", hw=((800, 1333),), residual_fp32=True, floor=False, family="t"):
    """Full-depth model at the benchmark geometry vs the fp32 oracle, gated by the committed operand floor (see FLOOR_RATIO_*).
    caption: 'short' = 81 tokens (NT = 2 VLFuse kernels), 'long' = 141 tokens (NT = 3), 'xlong' = 201 tokens (NT = 4).
    family: 't' = MQ-GLIP-T (BASELINE configs[1]), 'l' = MQ-GLIP-L (configs[3]: Swin-L 2-2-18-2, window 12, 8 fusion layers).
    floor: recompute the floor here instead of reading tests/golden/floor_bench.json (slow; tests/gpu_diag.py --ladder)."""
    from dataclasses import replace
    from oracle import detector as od, postprocess as opp
    from mq_det_amd.modeling import pipeline
    from mq_det_amd.modeling.query_selector import build_token_index
    from mq_det_amd.structures import ImageList
    spec, sd, cfg, model = _bench_model(dev, residual_fp32, family)
    P = model._plan
    B = len(hw)
    images, sizes, ids, am, pm, nv, bank = bench_inputs(spec, caption, hw)
    model.load_query_bank(bank)
    okey = ("bench_oracle", family, caption, tuple(hw))
    with torch.no_grad():
        if okey not in _CACHE:                         # the oracle does not depend on the product's precision switches
            _CACHE[okey] = od.forward(sd, spec, images, sizes, ids, am, pm, bank, return_intermediates=True)
        dets, inter = _CACHE[okey]
        ref_rows = oracle_rows(inter, am, pm, nv)
        fl = fl_rows = None
        if floor:
            from oracle.precision import RoundGemmOperands
            with RoundGemmOperands(H16):
                _, fl = od.forward(sd, spec, images, sizes, ids, am, pm, bank, return_intermediates=True)
            fl_rows = oracle_rows(fl, am, pm, nv)
        raw = model(ImageList(images.to(dev), sizes), captions=None, positive_map=pm, return_raw=True,
                    input_ids=ids.to(dev), attention_mask=am.to(dev))
        x = images.to(dev).to(H16).contiguous(memory_format=torch.channels_last)
        cg = pipeline.swin_forward(P, cfg, x)
    got = _bench_got(raw, cg, am, nv)
    case = bench_case_key(family, caption, hw)
    fixture = floor_fixture().get(case, {}) if residual_fp32 else {}
    tag = f"bench[{'MQ-GLIP-L,' if family == 'l' else ''}{caption},B={B}{'' if residual_fp32 else ',fp16 streams'}]"
    res = _bench_gate(tag, got, ref_rows, fixture, fl_rows, residual_fp32, case)
    h = inter["head"]
    # ---- detections, both score-aggregation widths of the boundary: LVIS-style
    # TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM = 3000 with 300 detections, and the default -1 (DYHEAD.NUM_CLASSES - 1) with 100
    labels = [k for k, v in pm.items() if len(v)]
    tokidx, label_ids = build_token_index(pm, labels, dev)
    for mode, (mdetr, ndet) in (("mdetr3000", (spec.mdetr_class_num, spec.detections_per_img)), ("dyhead", (-1, 100))):
        spec2 = replace(spec, mdetr_class_num=mdetr, detections_per_img=ndet)
        cfg.MODEL.ATSS.DETECTIONS_PER_IMG, prev = ndet, cfg.MODEL.ATSS.DETECTIONS_PER_IMG
        cfg.TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM, prev2 = mdetr, cfg.TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM
        try:
            with torch.no_grad():
                odets = opp.atss_postprocess(h["bbox_reg"], h["centerness"], h["dot_product_logits"], inter["anchors"], sizes, pm, spec2)
                post = pipeline.postprocess(cfg, raw["head"], raw["anchors"], sizes, tokidx, label_ids)
        finally:
            cfg.MODEL.ATSS.DETECTIONS_PER_IMG, cfg.TEST.MDETR_STYLE_AGGREGATE_CLASS_NUM = prev, prev2
        # LVIS mode keeps 300 detections: the oracle's best 100 must be reproduced.  The default mode keeps only 100, so its
        # "top 100" would be ALL detections including those sitting on the keep / drop cut-off (which any perturbation at the
        # floor level reshuffles): the better half (top 50) is required there.
        top = 100 if ndet >= 300 else 50
        for b in range(B):
            n = int(post["counts"][b])
            frac = _match_detections(post["boxes"][b, :n].cpu(), post["scores"][b, :n].cpu(), post["labels"][b, :n].cpu(),
                                     odets[b]["boxes"], odets[b]["scores"], odets[b]["labels"], top=top)
            need = 0.95 if H16 == torch.float16 else 0.85          # bf16: 8x the rounding step, see use_dtype
            r = {"name": f"{tag} {mode}: top-{top} detections matched (IoU>0.9, |ds|<0.03) img{b} "
                         f"n_hip={n} n_ref={len(odets[b]['boxes'])}", "max_err": 1 - frac, "mean_err": 0.0, "ref_absmax": 1.0,
                 "norm_err": 1 - frac, "tol": 1 - need, "ok": frac >= need}
            fxd = fixture.get(f"{mode}: detections img{b}")
            if fxd is not None:                        # bf16: the floor itself reshuffles a third of the list; the product may not be worse
                r["tol"] = max(r["tol"], fxd["norm"] + 0.05)
                r["ok"] = bool(r["norm_err"] <= r["tol"])
            if fl is not None:                         # what the operand floor itself reproduces
                fh = fl["head"]
                fdets = opp.atss_postprocess(fh["bbox_reg"], fh["centerness"], fh["dot_product_logits"], inter["anchors"], sizes, pm, spec2)
                r["floor_norm_err"] = 1 - _match_detections(fdets[b]["boxes"], fdets[b]["scores"], fdets[b]["labels"], odets[b]["boxes"],
                                                            odets[b]["scores"], odets[b]["labels"], top=top)
                r["floor_mean_err"] = 0.0
            elif fxd is not None:
                r["floor_norm_err"], r["floor_mean_err"] = fxd["norm"], 0.0
            res.append(r)
    _LADDER[(family, caption, B, residual_fp32)] = res
    return res


def check_benchmark_b8_graph(dev, B=8):
    """THE TIMED CONFIGURATION (bench.py / BASELINE configs[1]: B = 8 images 800x1333, 141-token caption, HIP-graph replay) against the
    oracle.  Image 0 of the batch is the image of the 'long, B = 1' case, so that case's oracle forward and its committed
    floor fixture apply to batch item 0 unchanged: (a) every stage row of item 0 of an eager B = 8 forward under the floor gate,
    (b) the detections the REPLAYED graph returns for item 0 against the oracle's (top-100, >= 95 %), (c) the replayed graph against the
    eager forward of the same batch for every image, (d) batch invariance: items 3 and 7 of the replayed batch against a B = 1 forward
    of that image alone (other grids, other key splits: same detections)."""
    from oracle import detector as od, postprocess as opp
    from mq_det_amd.modeling import pipeline
    from mq_det_amd.structures import ImageList
    spec, sd, cfg, model = _bench_model(dev)
    P = model._plan
    hw = ((800, 1333),)
    images1, sizes1, ids, am, pm, nv, bank = bench_inputs(spec, "long", hw)
    model.load_query_bank(bank)
    g = torch.Generator().manual_seed(70)
    images = torch.zeros(B, *images1.shape[1:])
    images[0] = images1[0]
    images[1:, :, :800, :1333] = torch.randn(B - 1, 3, 800, 1333, generator=g).to(H16).float()
    # items 3 and B - 1 are compared with B = 1 forwards of the same image (batch invariance): pure noise yields ~3 detections there (VERDICT r4
    # weak #3) -- give them the structured image of item 0 mirrored / upside down instead (hundreds of detections, other pixels than item 0)
    images[3, :, :800, :1333] = torch.flip(images1[0, :, :800, :1333], dims=[-1])
    images[B - 1, :, :800, :1333] = torch.flip(images1[0, :, :800, :1333], dims=[-2])
    sizes = [sizes1[0]] * B
    ids8, am8 = ids.expand(B, -1).contiguous(), am.expand(B, -1).contiguous()
    okey = ("bench_oracle", "t", "long", tuple(hw))
    kw = dict(captions=None, positive_map=pm, input_ids=ids8.to(dev), attention_mask=am8.to(dev))
    il = ImageList(images.to(dev), sizes)
    graph_was = model.use_hip_graph
    res = []
    try:
        with torch.no_grad():
            if okey not in _CACHE:
                _CACHE[okey] = od.forward(sd, spec, images1, sizes1, ids, am, pm, bank, return_intermediates=True)
            dets, inter = _CACHE[okey]
            ref_rows = oracle_rows(inter, am, pm, nv)
            model.use_hip_graph = False
            model.clear_caches()
            eager = model(il, **kw)
            raw = model(il, return_raw=True, **kw)
            cg = pipeline.swin_forward(P, cfg, images.to(dev).to(H16).contiguous(memory_format=torch.channels_last))
            model.use_hip_graph = True
            model.clear_caches()
            outs = [model(il, **kw) for _ in range(3)]
            captured = any(e.get("stage") == 2 for e in model._graphs.values())
            model.use_hip_graph = False
            singles = {b: model(ImageList(images[b:b + 1].to(dev), sizes[b:b + 1]), captions=None, positive_map=pm, input_ids=ids.to(dev),
                                attention_mask=am.to(dev))[0] for b in (3, B - 1)}
    finally:
        model.use_hip_graph = graph_was
    tag = f"bench[long,B={B},graph]"
    case = bench_case_key("t", "long", hw)
    res += _bench_gate(tag + " item 0:", _bench_got(raw, cg, am8, nv, item=0), ref_rows, floor_fixture().get(case, {}), None, True, case)
    res.append({"name": f"{tag} the third forward is a HIP-graph replay", "max_err": 0.0, "mean_err": 0.0, "ref_absmax": 1.0,
                "norm_err": 0.0 if captured else 1.0, "tol": 0.0, "ok": bool(captured)})

    def match(a, rb, rs, rl, what, top=100, need=0.95):
        frac = _match_detections(a.bbox.cpu(), a.get_field("scores").cpu(), a.get_field("labels").cpu(), rb, rs, rl, top=top)
        res.append({"name": f"{tag} {what} (top-{top}, IoU>0.9, |ds|<0.03) n={len(a)}", "max_err": 1 - frac, "mean_err": 0.0, "ref_absmax": 1.0,
                    "norm_err": 1 - frac, "tol": round(1 - need, 3), "ok": frac >= need})
    d0 = dets[0]
    match(outs[2][0], d0["boxes"], d0["scores"], d0["labels"], "replayed detections of item 0 vs the ORACLE")
    for b in range(B):
        e = eager[b]
        match(outs[2][b], e.bbox.cpu(), e.get_field("scores").cpu(), e.get_field("labels").cpu(), f"replay vs eager forward, item {b}", need=0.99)
    for b, sgl in singles.items():
        match(outs[2][b], sgl.bbox.cpu(), sgl.get_field("scores").cpu(), sgl.get_field("labels").cpu(), f"item {b} of the batch vs the B = 1 forward of that image")
    return res


# ----------------------------------------------------------------------------------------------------------------------
# vision-query extraction path
def check_roi_align(dev):
    """mq_roi_align_fwd vs the oracle (legacy + aligned, adaptive + fixed sampling, NHWC fp16 views and NCHW fp32) and both
    vs the reference's own RoIAlignForward CUDA kernel (ROIAlign_cuda.cu:16-123 compiled by oracle/build_ref.py)."""
    from oracle import roi as oroi, ref_native as rn
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(41)
    res = []
    N, C, H, W = 2, 256, 25, 42
    feat = torch.randn(N, C, H, W, generator=g)
    rois = torch.tensor([[0, 10., 20., 300., 190.], [1, 0., 0., 671., 399.], [0, 50., 50., 50.5, 50.2], [1, 600., 350., 700., 420.],
                         [0, -20., -10., 40., 30.], [1, 333.3, 111.1, 444.4, 222.2]])
    f16 = feat.to(H16)
    nhwc = f16.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)           # NHWC memory, NCHW view (the product's layout)
    for aligned in (False, True):
        for sr in (0, 2):
            ref = oroi.roi_align(f16.float(), rois, 7, 1.0 / 16, sr, aligned)
            got = ops.roi_align(nhwc, rois.to(dev), 7, 1.0 / 16, sr, aligned=aligned)
            res.append(_stat(f"roi_align aligned={aligned} sampling={sr}: NHWC fp16 view vs oracle", got, ref, tol=1e-5))
            got32 = ops.roi_align(feat.to(dev), rois.to(dev), 7, 1.0 / 16, sr, aligned=aligned)
            res.append(_stat(f"roi_align aligned={aligned} sampling={sr}: NCHW fp32 vs oracle", got32, oroi.roi_align(feat, rois, 7, 1.0 / 16, sr, aligned), tol=1e-5))
            gm = ops.roi_align(nhwc, rois.to(dev), 7, 1.0 / 16, sr, aligned=aligned, reduce_mean=True)
            res.append(_stat(f"roi_align aligned={aligned} sampling={sr}: fused bin mean", gm, got.mean((-1, -2)), tol=1e-5))
            if not aligned and PINS:
                pin = rn.roi_align(feat.to(dev), rois.to(dev), 7, 1.0 / 16, sr)
                res.append(_stat(f"PIN oracle.roi_align vs reference CUDA kernel: sampling={sr}", oroi.roi_align(feat, rois, 7, 1.0 / 16, sr, False), pin, tol=1e-5))
                res.append(_stat(f"PIN mq_roi_align_fwd vs reference CUDA kernel: sampling={sr}", got32, pin, tol=1e-5))
    # Pooler with boxes on several FPN levels (LevelMapper) vs the oracle pooler, NHWC fp16 pyramid
    from mq_det_amd.modeling.poolers import Pooler
    sizes = [(640, 800), (600, 720)]
    feats = [torch.randn(2, 256, -(-640 // s), -(-800 // s), generator=g).to(H16) for s in (8, 16, 32, 64, 128)]
    bl, tup = _query_targets(sizes, dev)
    scales = (0.125, 0.0625, 0.03125, 0.015625, 0.0078125)
    pl = Pooler((7, 7), scales, 0, use_v2=True)
    got = pl([f.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2) for f in feats], bl)
    ref = oroi.pooler([f.float() for f in feats], [t[0] for t in tup], 7, scales, 0)
    lv = pl.map_levels(bl)
    res.append(_stat(f"Pooler over FPN levels {sorted(set(lv.tolist()))} vs oracle pooler", got, ref, tol=1e-5))
    return res


def _query_targets(sizes, dev=None, seed=43):
    """A few labelled boxes per image as BoxLists (product) and plain tuples (oracle)."""
    from mq_det_amd.structures import BoxList
    g = torch.Generator().manual_seed(seed)
    bl, tup = [], []
    for (h, w) in sizes:
        n = 5
        xy = torch.rand(n, 2, generator=g) * torch.tensor([w * 0.6, h * 0.6])
        wh = torch.rand(n, 2, generator=g) * torch.tensor([w * 0.35, h * 0.35]) + 4
        boxes = torch.cat([xy, xy + wh], 1)
        boxes[0] = torch.tensor([1.0, 2.0, w - 2.0, h - 3.0])                       # a box that maps to a coarse level
        labels = torch.tensor([3, 1, 3, 2, 1])
        b = BoxList(boxes.clone() if dev is None else boxes.to(dev), (w, h), mode="xyxy")
        b.add_field("labels", labels if dev is None else labels.to(dev))
        bl.append(b)
        tup.append((boxes, labels, (w, h)))
    return bl, tup


def check_extract_query(dev):
    """GeneralizedVLRCNN_New.extract_query (ROIAlign HIP kernel + Pooler + bank update) vs the oracle's restatement, on the
    tiny model: from pixels (backbone run inside) and from features handed back by forward(return_backbone_features=True);
    level-selecting and all-level poolers; exclude_similar / max_query_number."""
    from collections import defaultdict
    from oracle import backbone as ob, roi as oroi
    from mq_det_amd.modeling.poolers import CustomPooler
    from mq_det_amd.structures import ImageList
    spec, sd, cfg, model, P = tiny(dev)
    images, sizes, ids, am, pm, bank = make_inputs(spec)
    with torch.no_grad():
        feats = ob.fpn_forward(sd, "backbone.fpn", ob.swin_forward(sd, "backbone.body", images, spec))
    bl, tup = _query_targets(sizes, dev)
    RB = cfg.MODEL.ROI_BOX_HEAD
    pool = (RB.POOLER_RESOLUTION, tuple(RB.POOLER_SCALES), RB.POOLER_SAMPLING_RATIO)
    res = []
    ref = oroi.extract_query(feats, tup, {}, pool, select_fpn_level=True, expand_ratio=cfg.VISION_QUERY.EXPAND_RATIO)
    got = model.extract_query(images=ImageList(images.to(dev), sizes), targets=bl, query_images=defaultdict(list))
    assert sorted(got) == sorted(ref) == [1, 2, 3]
    for lab in ref:
        res.append(_stat(f"extract_query (from pixels) label {lab} [n, 1, C]", got[lab], ref[lab], tol=4e-3))
    # second pass with exclude_similar on the same boxes: every candidate is a duplicate of a bank row -> bank unchanged
    got2 = model.extract_query(images=ImageList(images.to(dev), sizes), targets=bl, query_images={k: v.clone() for k, v in got.items()},
                               exclude_similar=True)
    same = all(len(got2[k]) == len(got[k]) for k in got)
    res.append({"name": "extract_query exclude_similar skips duplicates", "max_err": 0.0 if same else 1.0, "mean_err": 0.0, "ref_absmax": 1.0,
                "norm_err": 0.0 if same else 1.0, "tol": 0.0, "ok": same})
    got3 = model.extract_query(images=ImageList(images.to(dev), sizes), targets=bl, query_images=defaultdict(list), max_query_number=1)
    ok3 = all(len(v) == 1 for v in got3.values())
    res.append({"name": "extract_query max_query_number", "max_err": 0.0 if ok3 else 1.0, "mean_err": 0.0, "ref_absmax": 1.0,
                "norm_err": 0.0 if ok3 else 1.0, "tol": 0.0, "ok": ok3})
    # all-level pooler (VISION_QUERY.SELECT_FPN_LEVEL = False) on fp32 features handed over by the caller
    prev_pool, prev_flag = model.pooler, cfg.VISION_QUERY.SELECT_FPN_LEVEL
    try:
        cfg.VISION_QUERY.SELECT_FPN_LEVEL = False
        model.pooler = CustomPooler(output_size=(pool[0], pool[0]), scales=pool[1], sampling_ratio=pool[2], use_v2=True)
        ref5 = oroi.extract_query(feats, tup, {}, pool, select_fpn_level=False, expand_ratio=cfg.VISION_QUERY.EXPAND_RATIO)
        got5 = model.extract_query(targets=bl, query_images=defaultdict(list), visual_features=[f.to(dev) for f in feats], device=dev)
        for lab in ref5:
            res.append(_stat(f"extract_query (all levels, oracle features) label {lab} [n, 5, C]", got5[lab], ref5[lab], tol=1e-5))
    finally:
        model.pooler, cfg.VISION_QUERY.SELECT_FPN_LEVEL = prev_pool, prev_flag
    return res


def check_swin_mlp(dev, variants=None):
    """The fused Swin MLP half (LN prologue + fc1 + exact GELU + fc2 + residual + fused next LayerNorm in one kernel) vs a plain fp32
    statement on the same fp16-rounded weights: every supported width, ragged token counts, with / without delta / next-LN.
    variants: ("v2", flags) = mq_swin_mlp2_fwd with flags (bit 1 table GELU, bit 0 no pass / tail split, bit 2 everything through the tail
    kernel); default: erf / table / tail-only / unsplit.  The two long cases cross the pass / tail
    split on the device (C = 384: 33 600 tokens = 263 workgroups on 256 CUs; C = 192: 773 on 768 slots); 1030 / 777 / 562 tokens cross it on
    the emulator's 4-CU "chip" (tests/simt/include/hip/hip_runtime.h)."""
    from mq_det_amd import ops
    res = []
    variants = variants or (("v2", 0), ("v2", 2), ("v2", 4), ("v2", 1))
    for var in variants:
        g = torch.Generator().manual_seed(51)
        for C, M, use_delta, use_next in ((96, 1030, True, True), (96, 128, False, False), (192, 777, True, True), (384, 562, True, True),
                                          (384, 64, True, False), (96, 67200 * 2 + 5, True, True), (384, 33600, True, True),
                                          (192, 49452, True, True))[:5 if QUICK else 8]:
            if M > 30000 and var[1] & 4:
                continue
            x = torch.randn(M, C, generator=g) * 1.5
            delta = (torch.randn(M, C, generator=g) * 0.5).to(H16) if use_delta else None
            lg, lb = (torch.randn(C, generator=g) * 0.1 + 1).to(H16), (torch.randn(C, generator=g) * 0.1).to(H16)
            w1 = (torch.randn(4 * C, C, generator=g) / math.sqrt(C)).to(H16)
            b1 = (torch.randn(4 * C, generator=g) * 0.1).to(H16)
            w2 = (torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C)).to(H16)
            b2 = (torch.randn(C, generator=g) * 0.1).to(H16)
            ng, nb = (torch.randn(C, generator=g) * 0.1 + 1).to(H16), (torch.randn(C, generator=g) * 0.1).to(H16)
            xp = x + (delta.float() if use_delta else 0.0)
            h = F.layer_norm(xp, (C,), lg.float(), lb.float(), 1e-5).to(H16).float()          # the kernel feeds fp16 to the MFMAs
            hid = F.gelu(F.linear(h, w1.float(), b1.float())).to(H16).float()
            ref = xp + F.linear(hid, w2.float(), b2.float())
            nln = (ng.to(dev), nb.to(dev), 1e-5) if use_next else None
            dl = None if delta is None else delta.to(dev)
            w1f, w2f = ops.swin_mlp2_pack(w1, w2)
            r = ops.swin_mlp2(x.to(dev), dl, lg.to(dev), lb.to(dev), 1e-5, w1f.to(dev), b1.to(dev), w2f.to(dev), b2.to(dev), next_ln=nln,
                              flags=var[1])
            tag = (f"swin_mlp2[{'tail only' if var[1] & 4 else 'unsplit' if var[1] & 1 else 'split'},{'table' if var[1] & 2 else 'erf'}] "
                   f"C={C} M={M} delta={use_delta}")
            out, y = r if use_next else (r, None)
            if not (var[1] & 5) and use_next:
                # the two parts as two calls (flags bit 3 = main blocks only, bit 4 = tail blocks only; on the device they run on two
                # streams side by side): together bit for bit the single call
                a = ops.swin_mlp2(x.to(dev), dl, lg.to(dev), lb.to(dev), 1e-5, w1f.to(dev), b1.to(dev), w2f.to(dev), b2.to(dev), next_ln=nln,
                                  flags=var[1] | 8, into=(torch.full_like(out, float("nan")), torch.full_like(y, float("nan"))))
                a = ops.swin_mlp2(x.to(dev), dl, lg.to(dev), lb.to(dev), 1e-5, w1f.to(dev), b1.to(dev), w2f.to(dev), b2.to(dev), next_ln=nln,
                                  flags=var[1] | 16, into=a)
                res.append(_stat(f"{tag}: main-only + tail-only calls == one call (out)", a[0], out.float().cpu(), tol=0.0))
                res.append(_stat(f"{tag}: main-only + tail-only calls == one call (next LayerNorm)", a[1], y.float().cpu(), tol=0.0))
            res.append(_stat(f"{tag}: out (fp32 stream)", out, ref, tol=1e-3))
            if use_next:
                res.append(_stat(f"{tag}: fused next LayerNorm", y, F.layer_norm(ref, (C,), ng.float(), nb.float(), 1e-5), tol=2e-3))
    return res


# ----------------------------------------------------------------------------------------------------------------------
# MQ-GroundingDINO path: the multi-scale deformable attention operator (the model: tests/gdino_checks.py)
def check_msdeform_attn(dev, golden_dir=None):
    """mq_msdeform_attn_fwd vs (1) the reference-generated fixture tests/golden/msda.npz, (2) the reference's own CUDA kernel
    (ms_deform_im2col_cuda.cuh:237-299 via oracle/build_ref.py) -- which also pins oracle.gdino.ms_deform_attn_core --,
    (3) the oracle at the encoder's shape (4 levels of an 800x1344 image, Q = 22 323) with fp16 values; and the module
    forward (value / offset / weight projections, 2-d and 4-d reference points, padding mask) vs the oracle's restatement."""
    import os
    import numpy as np
    from oracle import gdino, ref_native as rn
    from mq_det_amd import ops
    from mq_det_amd.modeling import msdeform
    golden_dir = golden_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    gd = np.load(os.path.join(golden_dir, "msda.npz"))
    value, loc, attn = (torch.from_numpy(gd[k]) for k in ("value", "loc", "attn"))
    shapes = [tuple(s) for s in gd["shapes"].tolist()]
    res = []
    got = ops.ms_deform_attn(value.to(dev), shapes, loc.to(dev), attn.to(dev))
    res.append(_stat("msdeform fp32 vs reference fixture (multi_scale_deformable_attn_pytorch)", got, torch.from_numpy(gd["out"]), tol=1e-5))
    hw = torch.tensor(shapes, dtype=torch.int64)
    start = torch.cat([hw.new_zeros(1), (hw[:, 0] * hw[:, 1]).cumsum(0)[:-1]])
    if PINS:
        pin = rn.ms_deform_attn(value.to(dev), hw, start, loc.to(dev), attn.to(dev))
        res.append(_stat("PIN mq_msdeform_attn_fwd vs reference CUDA kernel", got, pin, tol=1e-5))
        res.append(_stat("PIN oracle.ms_deform_attn_core vs reference CUDA kernel", gdino.ms_deform_attn_core(value, shapes, loc, attn), pin, tol=1e-5))
    # encoder shape, fp16 values (the product's dtype), ragged query count
    g = torch.Generator().manual_seed(61)
    shapes = [(100, 168), (50, 84), (25, 42), (13, 21)]
    S = sum(h * w for h, w in shapes)
    v = torch.randn(1, S, 8, 32, generator=g).to(H16)
    Q = 2001
    loc = torch.rand(1, Q, 8, 4, 4, 2, generator=g) * 1.1 - 0.05
    attn = torch.rand(1, Q, 8, 16, generator=g).softmax(-1).reshape(1, Q, 8, 4, 4)
    ref = gdino.ms_deform_attn_core(v.float(), shapes, loc, attn)
    got = ops.ms_deform_attn(v.to(dev), shapes, loc.to(dev), attn.to(dev))
    res.append(_stat(f"msdeform fp16 values, 4 levels of 800x1344, Q={Q}", got, ref, tol=1e-3))
    # module forward
    sd = gdino.make_msda_weights(prefix="attn")
    W = msdeform.pack_msda(sd, "attn", dev, dtype=H16)
    sdh = {k: t.to(H16).float() for k, t in sd.items()}
    shapes = [(20, 24), (10, 12), (5, 6), (3, 3)]
    S = sum(h * w for h, w in shapes)
    x = torch.randn(2, S, 256, generator=g).to(H16)
    mask = torch.zeros(2, S, dtype=torch.bool)
    mask[1, S - 40:] = True
    for nd in (2, 4):
        rp = torch.rand(2, S, 4, nd, generator=g)
        if nd == 4:
            rp[..., 2:] = rp[..., 2:] * 0.3 + 0.05
        ref = gdino.ms_deform_attn(sdh, "attn", x.float(), None, rp, shapes, key_padding_mask=mask)
        got = msdeform.ms_deform_attn(W, x.to(dev), None, rp.to(dev), shapes, key_padding_mask=mask.to(dev))
        res.append(_stat(f"MultiScaleDeformableAttention.forward reference_points[..., {nd}]", got, ref, tol=6e-3))
    return res
