"""The gfx950 kernel SOURCES (mq_det_amd/csrc/*.hip), executed lane by lane on the host, against the oracle -- without a GPU.

tests/simt/ compiles the .hip files unchanged for x86-64 against a stand-in HIP runtime in which every thread is a fiber: a
wavefront's 64 lanes meet at each MFMA / shuffle / ballot / LDS transpose read, a workgroup meets at __syncthreads().  The
same checks that run on the MI355X (tests/parity_checks.py, tests/gdino_checks.py: the product's torch wrappers -> C ABI ->
kernels, compared with the oracle / plain fp32 restatements) run here on CPU tensors, so a wrong fragment layout, LDS image,
index or mask in a kernel shows up in the CPU suite -- before any GPU time is spent.  What this cannot see: timing, occupancy,
register spills, and the last bits of expf / MFMA rounding order; the `-m gpu` suite stays the parity gate.

TEST INFRASTRUCTURE ONLY: the emulation library is loaded here and nowhere else; mq_det_amd/ raises without a GPU."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

CPU = torch.device("cpu")

# the emulation library is built with the host clang of the ROCm toolchain (tests/simt/build_emu.py); an image without it cannot run
# these tests at all (a build that FAILS with the compiler present is an error, not a skip)
_CXX = os.environ.get("SIMT_CXX", "/opt/rocm/lib/llvm/bin/clang++")
pytestmark = pytest.mark.skipif(not os.path.exists(_CXX), reason=f"{_CXX} not found: the kernel-source emulation cannot be built here")


@pytest.fixture(scope="module")
def kernels():
    import simt
    import parity_checks as pc
    from mq_det_amd.modeling import detector, gdino, pipeline, gdino_pipeline as gp

    def prepare(self, device=None):
        self._validate_config()
        self._plan = pipeline.build_plan(self.state_dict(), self.cfg, CPU, dtype=detector.compute_dtype(self.cfg))
        self._plan_key, self.use_hip_graph = CPU, False
        return self._plan

    def prepare_gdino(self, device=None):
        self._plan = gp.build_gdino_plan(self.state_dict(), self.cfg, CPU, self._swin, dtype=detector.compute_dtype(self.cfg))
        self._plan_key, self.use_hip_graph = CPU, False
        return self._plan

    saved = (detector.GeneralizedVLRCNN_New.prepare, gdino.GroundingDINO.prepare, pc.QUICK, pc.PINS, dict(pc._CACHE))
    detector.GeneralizedVLRCNN_New.prepare, gdino.GroundingDINO.prepare = prepare, prepare_gdino
    pc.QUICK, pc.PINS = os.environ.get("MQ_SIMT_FULL", "0") != "1", False
    import gdino_checks as gc
    saved_gc = dict(gc._CACHE)
    pc._CACHE.clear()
    gc._CACHE.clear()                  # models cached by other test modules carry plans built by THEIR prepare() stand-ins
    with simt.installed():
        yield pc
    detector.GeneralizedVLRCNN_New.prepare, gdino.GroundingDINO.prepare, pc.QUICK, pc.PINS = saved[:4]
    pc._CACHE.clear()
    pc._CACHE.update(saved[4])
    gc._CACHE.clear()
    gc._CACHE.update(saved_gc)


def _assert_ok(results):
    results = results if isinstance(results, list) else [results]
    assert results, "no results"
    bad = [f"{r['name']}: norm_err {r['norm_err']:.2e} > tol {r['tol']:.1e}" for r in results if not r["ok"]]
    assert not bad, "\n".join(bad)


def test_emulation_library_exports_the_whole_c_abi():
    import simt
    from mq_det_amd import ops
    lib = simt.library()
    assert lib.mq_abi_version() > 0 and all(hasattr(lib, n) for n in ops.EXPORTS)
    # the product binding is untouched by loading it
    assert ops._LIB is None or ops._LIB is not lib


def test_emulation_unit_kernels():
    """The emulation's own semantics: one MFMA against a plain matmul in the documented fragment layout, shuffles / ballot, a
    deliberate LDS race that the schedule permutations expose (and the barrier-ed version they do not), a divergent barrier that
    is reported as a failed launch instead of hanging."""
    import ctypes
    import simt
    lib = simt.library()
    vp = ctypes.c_void_p
    g = torch.Generator().manual_seed(0)
    A, Bt = torch.randn(16, 32, generator=g).half(), torch.randn(16, 32, generator=g).half()
    C = torch.zeros(16, 16)
    assert lib.simt_selftest_mfma(vp(A.data_ptr()), vp(Bt.data_ptr()), vp(C.data_ptr())) == 0
    assert torch.allclose(C, A.float() @ Bt.float().t(), atol=1e-5)
    A, Bt = torch.randn(32, 16, generator=g).half(), torch.randn(32, 16, generator=g).half()
    C = torch.zeros(32, 32)
    assert lib.simt_selftest_mfma32(vp(A.data_ptr()), vp(Bt.data_ptr()), vp(C.data_ptr())) == 0
    assert torch.allclose(C, A.float() @ Bt.float().t(), atol=1e-5)
    x, y, m = torch.randn(64, generator=g), torch.zeros(64), torch.zeros(1, dtype=torch.int64)
    assert lib.simt_selftest_shfl(vp(x.data_ptr()), vp(y.data_ptr()), vp(m.data_ptr())) == 0
    idx = torch.arange(64)
    assert torch.equal(y, x[idx ^ 1] + x[(idx + 5) & 63])
    assert int(m) & (2 ** 64 - 1) == sum(1 << i for i in range(64) if x[i] > 0) or int(m) == sum(1 << i for i in range(64) if x[i] > 0) - 2 ** 64
    seen = {}
    for barrier in (1, 0):
        outs = set()
        for mode, seed in (("ascending", 0), ("descending", 0), ("random", 1), ("random", 2)):
            simt.set_schedule(mode, seed)
            o = torch.full((64,), -7, dtype=torch.int32)
            assert lib.simt_selftest_race(vp(o.data_ptr()), barrier) == 0
            outs.add(tuple(o.tolist()))
        seen[barrier] = outs
    simt.set_schedule("ascending")
    assert seen[1] == {tuple(range(64))}                       # race-free: one answer under every schedule
    assert len(seen[0]) > 1                                    # missing barrier: the schedules disagree
    o = torch.zeros(64, dtype=torch.int32)
    assert lib.simt_selftest_deadlock(vp(o.data_ptr())) != 0   # hipErrorLaunchFailure instead of a hang


@pytest.mark.parametrize("schedule,names", [
    (("descending", 0), ("check_attention_strided", "check_window_attention", "check_gcp_block", "check_layernorm", "check_swin_mlp",
                         "check_nms", "check_post_golden", "check_msdeform_attn")),
    (("random", 1), ("check_dcn", "check_vlfuse_kernels", "check_conv3x3", "check_post_fused")),
    (("descending", 0), ("check_post_fused", "check_bert_attn_qkv", "check_gcp_attn_fused")),
    (("random", 3), ("check_bert_attn_qkv", "check_gcp_attn_fused"))])
def test_kernels_are_insensitive_to_the_wave_schedule(kernels, schedule, names):
    """Race check of the shipped kernels: the same parity checks with the fibers resumed in descending / pseudo-random order
    (a consumer wave then runs before its producer unless a barrier orders them).  MQ_SIMT_FULL=1: every check under both."""
    import simt
    simt.set_schedule(*schedule)
    try:
        for name in names:
            _assert_ok(getattr(kernels, name)(CPU))
    finally:
        simt.set_schedule("ascending")


@pytest.mark.parametrize("cfg", [
    dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True),
    dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True, clamp=50000.0, big=True),
    dict(B=1, H=8, D=32, Nq=200, Nk=1333, nsplit=4),
    dict(B=1, H=8, D=32, Nq=37, Nk=61),
    dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True, kvlen=True),
])
def test_attention_kernel(kernels, cfg):
    _assert_ok(kernels.check_attention(CPU, **cfg))


def test_attention_strided_views(kernels):
    _assert_ok(kernels.check_attention_strided(CPU))


@pytest.mark.parametrize("name", ["check_window_attention", "check_gcp_block", "check_pre_select", "check_vlfuse_kernels", "check_vl_fuse",
                                  "check_dcn", "check_dyconv", "check_post_golden", "check_score_agg", "check_layernorm", "check_nms", "check_swin_mlp",
                                  "check_conv3x3", "check_roi_align", "check_msdeform_attn", "check_swin_fpn", "check_align_fused", "check_post_fused", "check_attention_text", "check_bert_attn_qkv", "check_gcp_attn_fused", "check_patch_embed", "check_bert_clamp_fused"])
def test_kernel_block(kernels, name):
    _assert_ok(getattr(kernels, name)(CPU))


def test_dcn_barrier_schedules_are_equal():
    """MQ_DCN_SYNC: one barrier per k-step (default since round 3) and the former two produce EQUAL outputs (same MFMA order) -- the switch
    is read once per process, so each schedule runs in a process of its own; the descending / random wave orders of
    test_kernels_are_insensitive_to_the_wave_schedule cover the default."""
    import subprocess
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); import simt; from mq_det_amd import ops\n"
            "g = torch.Generator().manual_seed(5)\n"
            "x = torch.randn(2, 19, 23, 256, generator=g).half(); om = torch.randn(2, 27, 19, 23, generator=g) * 1.5\n"
            "w = (torch.randn(256, 9 * 256, generator=g) / 48).half(); b = torch.zeros(256).half()\n"
            "with simt.installed():\n    y = ops.dcnv2(x, om, w, b, 1)[0]\n"
            "torch.save(y, sys.argv[1])\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    outs = []
    with tempfile.TemporaryDirectory() as tmp:
        for sync in ("1", "2"):
            f = os.path.join(tmp, f"y{sync}.pt")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, MQ_DCN_SYNC=sync), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-400:]
            outs.append(torch.load(f))
    assert torch.equal(outs[0], outs[1]) and float(outs[0].float().abs().mean()) > 0.01


def test_fpn_topdown_fused_equals_interpolate_plus_add(kernels, monkeypatch):
    """mq_add_upsample_nearest (KERNELS["FPN_TOPDOWN_FUSED"] = 1) against F.interpolate(mode="nearest")
    + add on even and odd size pairs (the reference's top-down step, fpn.py:82-95): EQUAL outputs; and the Swin + FPN check with it on."""
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(12)
    for dt in (torch.float16, torch.bfloat16):
        for (H, W), (Hc, Wc) in (((100, 168), (50, 84)), ((25, 42), (13, 21)), ((13, 21), (7, 11)), ((7, 9), (4, 5)), ((5, 5), (1, 1))):
            lat = torch.randn(2, H, W, 256, generator=g).to(dt)
            coarse = torch.randn(2, Hc, Wc, 256, generator=g).to(dt)
            ref = (lat.float() + F.interpolate(coarse.permute(0, 3, 1, 2).float(), size=(H, W), mode="nearest").permute(0, 2, 3, 1)).to(dt)
            got = ops.add_upsample_nearest_(lat.clone(), coarse)
            assert torch.equal(got, ref), (dt, H, W, Hc, Wc)
    monkeypatch.setenv("MQ_FPN_TOPDOWN_FUSED", "1")
    _assert_ok(kernels.check_swin_fpn(CPU))


def test_pooled_tokens_fused_equals_avg_pool_and_cat(kernels, monkeypatch):
    """mq_pool2x2_tokens_fwd (KERNELS["POOLED_TOKENS_FUSED"] = 1) against the reference's statement (generalized_vl_rcnn_new.py:291-293: five
    F.avg_pool2d(f, 2) + concat over the tokens) on even and odd level sizes, batch-strided inputs included: EQUAL outputs; and the full-model
    check with it on."""
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(13)
    for dt in (torch.float16, torch.bfloat16):
        for sizes in (((100, 168), (50, 84), (25, 42), (13, 21), (7, 11)), ((9, 7), (5, 4), (3, 2)), ((2, 2),)):
            big = [torch.randn(3, h, w, 256, generator=g).to(dt) for h, w in sizes]
            feats = [x[:2].permute(0, 3, 1, 2) for x in big]                       # [B,C,H,W] views of NHWC storage, as the FPN returns them
            ref = torch.cat([F.avg_pool2d(f.float(), 2).to(dt).permute(0, 2, 3, 1).flatten(1, 2) for f in feats], 1)
            got = ops.pool2x2_tokens(feats)
            assert got.shape == ref.shape and torch.equal(got, ref), (dt, sizes)
    monkeypatch.setenv("MQ_POOLED_TOKENS_FUSED", "1")
    _assert_ok(kernels.check_full_model(CPU))


@pytest.mark.parametrize("clamp", [False, True])
def test_bert_layer(kernels, clamp):
    _assert_ok(kernels.check_bert_layer(CPU, clamp))


def test_full_model_smoke_check(kernels):
    """exactly what __graft_entry__.smoke() runs on cuda:0: tiny-depth MQ-GLIP-T, every stage + detections against the oracle"""
    _assert_ok(kernels.check_full_model(CPU))


@pytest.mark.parametrize("name", ["check_attention_qk_mask", "check_msdeform_attn_q", "check_gdino_state_dict_and_quirks"])
def test_groundingdino(kernels, name):
    import gdino_checks as gc
    _assert_ok(_no_graph_rows(getattr(gc, name)(CPU)))


def _no_graph_rows(results):
    """rows that assert an actual HIP-graph capture have no meaning without a device"""
    return [r for r in results if "HIP-graph" not in r["name"]]


@pytest.mark.parametrize("name", ["check_extract_query", "check_gdino_tiny", "large",
                                  pytest.param("check_vlfuse_heads_mask", marks=pytest.mark.skipif(
                                      os.environ.get("MQ_SIMT_FULL", "0") != "1", reason="a minute on the emulator: MQ_SIMT_FULL=1"))])
def test_model_blocks(kernels, name):
    """vision-query extraction (ROIAlign + poolers), the tiny MQ-GroundingDINO model stage by stage, Swin-L (window 12) blocks"""
    import gdino_checks as gc
    if name == "large":
        _assert_ok(kernels.check_window_attention(CPU, large=True) + kernels.check_swin_fpn(CPU, large=True))
    else:
        _assert_ok(_no_graph_rows(getattr(kernels if hasattr(kernels, name) else gc, name)(CPU)))


# ---- bf16 operands: the *_bf16 entry points (the same kernel sources compiled with -DMQ_BF16, BASELINE.json configs[3])
@pytest.fixture()
def bf16(kernels):
    kernels.use_dtype(torch.bfloat16)
    yield kernels
    kernels.use_dtype(torch.float16)


@pytest.mark.parametrize("name", ["check_window_attention", "check_gcp_block", "check_pre_select", "check_vlfuse_kernels", "check_dcn",
                                  "check_post_golden", "check_layernorm", "check_swin_mlp", "check_conv3x3", "check_msdeform_attn",
                                  "check_attention_strided", "check_attention_text", "check_bert_attn_qkv", "check_gcp_attn_fused", "check_patch_embed", "check_bert_clamp_fused"])
def test_bf16_kernel_block(bf16, name):
    res = getattr(bf16, name)(CPU)
    _assert_ok(res)
    assert all(r["name"].startswith("[bf16]") or r["tol"] == 0.0 or "count" in r["name"] or "(exact)" in r["name"] or "(px)" in r["name"]
               for r in (res if isinstance(res, list) else [res]))


def test_bf16_full_model(bf16):
    """MODEL.COMPUTE_DTYPE = "bfloat16": the tiny-depth MQ-GLIP-T forward on the bf16 kernels, every stage within 8x the fp16 tolerance"""
    _assert_ok(bf16.check_full_model(CPU))


def test_mixed_16bit_operands_are_rejected(kernels):
    from mq_det_amd import ops
    q = torch.zeros(1, 8, 64, dtype=torch.float16)
    with pytest.raises(TypeError, match="fp16 and bf16"):
        ops._fn(ops._LIB, "mq_attn_fwd", q, q.to(torch.bfloat16))


# ---- the resident-key attention kernel for text-sized key sequences (csrc/attn_resident.hip, opt-in: MQ_ATTN_RESIDENT=1)
@pytest.fixture()
def resident(kernels, monkeypatch):
    monkeypatch.setenv("MQ_ATTN_RESIDENT", "1")
    from mq_det_amd import ops
    calls = []
    orig = ops._fn

    def spy(lib, name, *ts):
        calls.append(name)
        return orig(lib, name, *ts)
    monkeypatch.setattr(ops, "_fn", spy)
    yield kernels, calls


@pytest.mark.parametrize("cfg", [
    dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True),
    dict(B=2, H=12, D=64, Nq=256, Nk=256, mask=True, clamp=50000.0, big=True),
    dict(B=1, H=8, D=32, Nq=37, Nk=61),
    dict(B=2, H=8, D=32, Nq=1, Nk=9),
    dict(B=3, H=2, D=64, Nq=130, Nk=141, mask=True, kvlen=True),
    dict(B=1, H=4, D=32, Nq=900, Nk=200),
])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_resident_attention_kernel(resident, cfg, dtype):
    pc, calls = resident
    pc.use_dtype(dtype)
    try:
        _assert_ok(pc.check_attention(CPU, **cfg))
    finally:
        pc.use_dtype(torch.float16)
    assert "mq_attn_resident_fwd" in calls and "mq_attn_fwd" not in calls


def test_resident_attention_masks_views_and_schedules(resident):
    import gdino_checks as gc
    import simt
    pc, calls = resident
    _assert_ok(pc.check_attention_strided(CPU))
    _assert_ok(gc.check_attention_qk_mask(CPU))            # byte masks (shared / per head), T = 100, strided K / V^T slices, kv_len
    assert calls.count("mq_attn_resident_fwd") >= 5
    for mode in (("descending", 0), ("random", 3)):
        simt.set_schedule(*mode)
        try:
            _assert_ok(pc.check_attention(CPU, B=2, H=3, D=64, Nq=200, Nk=256, mask=True, kvlen=True))
            _assert_ok(pc.check_bert_layer(CPU, True))
        finally:
            simt.set_schedule("ascending")
    # long key sequences and key splits: the chunked S^T kernel; a per-(query, key) mask over a long sequence: mq_attn_fwd
    del calls[:]
    _assert_ok(pc.check_attention(CPU, B=1, H=8, D=32, Nq=64, Nk=1333, nsplit=4))
    assert calls == ["mq_attn_chunked_fwd"]


@pytest.mark.parametrize("cfg", [
    dict(B=1, H=8, D=32, Nq=200, Nk=5577, nsplit=4),                                   # the GCP pre-select shape, split over keys
    dict(B=2, H=8, D=32, Nq=70, Nk=700, mask=True, nsplit=2),
    dict(B=2, H=2, D=64, Nq=130, Nk=600, mask=True, kvlen=True, clamp=50000.0, big=True),
    dict(B=1, H=8, D=32, Nq=900, Nk=900),                                              # MQ-GroundingDINO decoder self-attention
    dict(B=1, H=2, D=32, Nq=37, Nk=257, nsplit=2),                                     # second chunk holds one key
    dict(B=1, H=2, D=32, Nq=37, Nk=100, nsplit=3),                                     # splits without any chunk
])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_chunked_attention_kernel(resident, cfg, dtype):
    pc, calls = resident
    pc.use_dtype(dtype)
    try:
        _assert_ok(pc.check_attention(CPU, **cfg))
    finally:
        pc.use_dtype(torch.float16)
    assert calls == ["mq_attn_chunked_fwd"]


def test_chunked_attention_pre_select_and_schedules(resident):
    import simt
    pc, calls = resident
    _assert_ok(pc.check_pre_select(CPU))
    assert "mq_attn_chunked_fwd" in calls
    for mode in (("descending", 0), ("random", 5)):
        simt.set_schedule(*mode)
        try:
            _assert_ok(pc.check_attention(CPU, B=1, H=4, D=32, Nq=200, Nk=1400, nsplit=2))     # 6 chunks: the LDS double buffer wraps twice
            _assert_ok(pc.check_attention(CPU, B=1, H=2, D=64, Nq=100, Nk=800, mask=True))
        finally:
            simt.set_schedule("ascending")


@pytest.mark.skipif(os.environ.get("MQ_SIMT_FULL", "0") != "1", reason="MQ_SIMT_FULL=1 (the BERT layer, pre-select and attention checks above cover the kernels)")
def test_resident_attention_full_model(resident):
    """every text-sized attention of the tiny MQ-GLIP-T forward (BERT layers, VLDyHead BERT copies with the clamp) on the resident
    kernel: the smoke() check end to end"""
    pc, calls = resident
    pc._CACHE.clear()
    _assert_ok(pc.check_full_model(CPU))
    assert calls.count("mq_attn_resident_fwd") >= 4


# ---- the load-batched LayerNorm (csrc/layernorm2.hip, opt-in: MQ_LN_VARIANT=2)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_layernorm2_is_bit_identical_to_layernorm(kernels, monkeypatch, dtype):
    """every width class (one .. six chunks per lane), every stream combination (fp16 / fp32 input, residual, second output, sum),
    ragged row counts incl. fewer rows than one pass and the 64-rows-per-block regime: the outputs of the two kernels are EQUAL"""
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(7)
    for rows, C in ((1000, 96), (5, 192), (777, 256), (130, 384), (65, 768), (50, 1536), (9, 2048), (7, 3072), (64 * 2048 + 3, 96)):
        x32 = torch.randn(rows, C, generator=g) * 2 + 0.5
        r32 = torch.randn(rows, C, generator=g)
        w, b = (torch.randn(C, generator=g) * 0.1 + 1).to(dtype), (torch.randn(C, generator=g) * 0.1).to(dtype)
        for x, res in ((x32.to(dtype), None), (x32, None), (x32.to(dtype), r32.to(dtype)), (x32, r32.to(dtype)), (x32.to(dtype), r32), (x32, r32)):
            if rows > 100000 and (res is None or x.dtype != torch.float32):
                continue                                     # the large case once (fp32 stream + residual: the Swin stage-1 call)
            outs = {}
            for variant in ("1", "2"):
                monkeypatch.setenv("MQ_LN_VARIANT", variant)
                outs[variant] = ops.layer_norm(x, w, b, 1e-5, residual=res, want_sum=True, want_y32=True)
            a, c = outs["1"], outs["2"]
            a, c = (a if isinstance(a, tuple) else (a,)), (c if isinstance(c, tuple) else (c,))
            assert len(a) == len(c) == (3 if res is not None else 2)
            for t1, t2 in zip(a, c):
                assert t1.dtype == t2.dtype and torch.equal(t1, t2), (rows, C, x.dtype, None if res is None else res.dtype)


def test_layernorm2_through_the_model(kernels, monkeypatch):
    monkeypatch.setenv("MQ_LN_VARIANT", "2")
    kernels._CACHE.clear()
    _assert_ok(kernels.check_layernorm(CPU))
    _assert_ok(kernels.check_bert_layer(CPU, True))
    _assert_ok(kernels.check_swin_fpn(CPU))


# ---- the offset conv with unconditional in-flight loads (csrc/conv_small2.hip, opt-in: MQ_OFFSET_CONV_VARIANT=2)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_offset_conv_v2_equals_v1(kernels, monkeypatch, dtype):
    """patches at every image border, ragged sizes smaller than a patch, 1 / 2 channel passes (C = 64, 128, 256), level views of a
    token buffer: the two kernels' outputs are EQUAL; plus the DyConv block on it"""
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(3)
    for B, H, W, C in ((2, 13, 21, 256), (1, 7, 11, 256), (1, 3, 5, 128), (2, 17, 33, 64), (1, 25, 42, 256)):
        x = torch.randn(B, H, W, C, generator=g).to(dtype)
        w = torch.zeros(32, 9 * C, dtype=dtype)
        w[:27] = (torch.randn(27, 9 * C, generator=g) / 48).to(dtype)
        bias = torch.randn(27, generator=g).to(dtype)
        outs = {}
        for variant in ("1", "2"):
            monkeypatch.setenv("MQ_OFFSET_CONV_VARIANT", variant)
            outs[variant] = ops.conv3x3_nchw32(x, w, bias, 27)
        assert torch.equal(outs["1"], outs["2"]), (B, H, W, C)
        big = torch.zeros(B, H * W + 37, C, dtype=dtype)                       # a pyramid level inside a larger token buffer
        big[:, 5:5 + H * W] = x.reshape(B, H * W, C)
        assert torch.equal(ops.conv3x3_nchw32(big[:, 5:5 + H * W].reshape(B, H, W, C), w, bias, 27), outs["1"])
    monkeypatch.setenv("MQ_OFFSET_CONV_VARIANT", "2")
    kernels.use_dtype(dtype)
    try:
        _assert_ok(kernels.check_conv3x3(CPU))
        if dtype == torch.float16:
            _assert_ok(kernels.check_dyconv(CPU))
    finally:
        kernels.use_dtype(torch.float16)


# ---- Swin PatchMerging gather + LayerNorm in one kernel (csrc/layernorm2.hip, opt-in: MQ_PATCH_MERGE_FUSED=1)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_patch_merge_ln_equals_cat_plus_layernorm(kernels, monkeypatch, dtype):
    """even / odd H and W (zero padding), every Swin width (C = 96 .. 768: 4C up to 3072), fp32 and 16-bit input: EQUAL to
    F.pad + four strided slices + cat + mq_layernorm_fwd; then the Swin + FPN check with the switch on"""
    import torch.nn.functional as F
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(11)
    for B, H, W, C in ((2, 8, 12, 96), (1, 7, 11, 96), (2, 5, 6, 192), (1, 9, 4, 384), (1, 3, 3, 768), (1, 1, 1, 96)):
        for xd in (torch.float32, dtype):
            x = (torch.randn(B, H, W, C, generator=g) * 2 + 0.3).to(xd)
            w, b = (torch.randn(4 * C, generator=g) * 0.1 + 1).to(dtype), (torch.randn(4 * C, generator=g) * 0.1).to(dtype)
            y = F.pad(x, (0, 0, 0, W % 2, 0, H % 2)) if (H % 2 or W % 2) else x
            y = torch.cat([y[:, 0::2, 0::2], y[:, 1::2, 0::2], y[:, 0::2, 1::2], y[:, 1::2, 1::2]], -1)
            ref = ops.layer_norm(y.reshape(B, -1, 4 * C).contiguous(), w, b, 1e-5)
            got = ops.patch_merge_ln(x, w, b, 1e-5)
            assert got.shape == ref.shape and got.dtype == ref.dtype and torch.equal(got, ref), (B, H, W, C, xd)
    monkeypatch.setenv("MQ_PATCH_MERGE_FUSED", "1")
    kernels.use_dtype(dtype)
    kernels._CACHE.clear()
    try:
        _assert_ok(kernels.check_swin_fpn(CPU))
    finally:
        kernels.use_dtype(torch.float16)
        kernels._CACHE.clear()


# ---- the offset conv of all pyramid levels in one launch, weights in registers (csrc/conv_small3.hip, MQ_OFFSET_CONV_VARIANT=3: the default since round 4)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_offset_conv_group_kernel(kernels, monkeypatch, dtype):
    """every level of a pyramid from one launch: against F.conv2d and against the per-level kernel (parity_checks.check_conv3x3_group:
    tile-edge sizes, runs of tiles that cross image / level boundaries), then the DyConv block and the tiny model's head through it;
    the ascending / descending / random wave orders of the emulator give the same bits (the exchange buffer aliases the window: barriers)"""
    from mq_det_amd import ops
    monkeypatch.setenv("MQ_OFFSET_CONV_VARIANT", "3")
    assert ops.KERNELS["OFFSET_CONV_VARIANT"] == 3
    kernels.use_dtype(dtype)
    calls = []
    real = ops.conv3x3_nchw32_group
    monkeypatch.setattr(ops, "conv3x3_nchw32_group", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    try:
        _assert_ok(kernels.check_conv3x3_group(CPU))
        n0 = len(calls)
        _assert_ok(kernels.check_dyconv(CPU))
        assert len(calls) > n0, "the DyConv layer did not take the grouped offset conv"
    finally:
        kernels.use_dtype(torch.float16)
    if dtype == torch.float16:
        g = torch.Generator().manual_seed(8)
        sizes = [(9, 17), (5, 9), (3, 5)]
        lv = [torch.randn(2, h, w, 256, generator=g).half() for h, w in sizes]
        wp = torch.zeros(32, 9 * 256, dtype=torch.float16)
        wp[:27] = (torch.randn(27, 9 * 256, generator=g) / 48).half()
        bias = torch.randn(27, generator=g).half()
        import simt
        base = real(lv, wp, bias, 27)
        try:
            for sched in (("descending", 0), ("random", 5), ("random", 11)):        # wave orders inside a workgroup: a missing barrier shows
                simt.set_schedule(*sched)
                for a, b_ in zip(base, real(lv, wp, bias, 27)):
                    assert torch.equal(a, b_), sched
        finally:
            simt.set_schedule("ascending")


# ---- the DyConv epilogue of all levels in two launches (csrc/dyconv.hip, opt-in: MQ_DYCONV_EPILOGUE_GROUPED=1)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_dyconv_epilogue_group(kernels, monkeypatch, dtype):
    """grouped == per-level launches bit for bit (1 .. 5 levels, every branch mix; random pyramids: test_simt_fuzz_cpu.py), then the DyConv block
    with the switch on (the tiny model on it: tools/simt_checks.py with MQ_DYCONV_EPILOGUE_GROUPED=1, and the isolated GPU body)"""
    from mq_det_amd import ops
    kernels.use_dtype(dtype)
    try:
        _assert_ok(kernels.check_dyconv_epilogue_group(CPU))
        monkeypatch.setenv("MQ_DYCONV_EPILOGUE_GROUPED", "1")
        assert ops.KERNELS["DYCONV_EPILOGUE_GROUPED"] == 1
        calls = []
        real = ops.dyconv_epilogue_group
        monkeypatch.setattr(ops, "dyconv_epilogue_group", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        if dtype == torch.float16:
            _assert_ok(kernels.check_dyconv(CPU))
            assert calls, "the DyConv layer did not take the grouped epilogue"
    finally:
        kernels.use_dtype(torch.float16)


# ---- out-of-bounds check: every library argument against a guard page (tests/simt/guard.py), in a subprocess
def _oob(mode, names, env=None, timeout=1200):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "simt", "oob_check.py"), mode, *names], capture_output=True, text=True,
                       timeout=timeout, env=dict(os.environ, **(env or {})))
    return r.returncode, [ln for ln in r.stdout.splitlines() if ln.startswith(("OK", "MISMATCH"))], r.stderr[-600:]


OPT_IN = {"MQ_ATTN_RESIDENT": "1", "MQ_LN_VARIANT": "2", "MQ_OFFSET_CONV_VARIANT": "2", "MQ_PATCH_MERGE_FUSED": "1", "MQ_FPN_VIA_DCN": "1",
          "MQ_NMS_EARLY_STOP": "1"}


@pytest.mark.parametrize("mode", ["end", "start"])
def test_no_kernel_touches_memory_outside_its_buffers(mode):
    """Inputs, outputs and workspaces of every call sit directly against a PROT_NONE page (after them: mode "end", before: "start"); one
    byte too far is a SIGSEGV.  Shipped kernels on a cross-section of checks, and the opt-in kernels of section 12 (which have never
    run on a device, where such an access is a silent wrong read or a memory fault that takes the process down)."""
    names = ["attention_small", "check_layernorm", "check_window_attention", "check_conv3x3", "check_bert_attn_qkv", "check_gcp_attn_fused"]      # (all 20 check groups pass: MQ_SIMT_FULL=1)
    if os.environ.get("MQ_SIMT_FULL", "0") == "1":
        names += ["check_gcp_block", "check_pre_select", "check_vlfuse_kernels", "check_vl_fuse", "check_dcn", "check_dyconv", "check_post_golden",
                  "check_score_agg", "check_nms", "check_swin_mlp", "check_roi_align", "check_msdeform_attn", "check_swin_fpn", "check_attention_qk_mask",
                  "check_msdeform_attn_q", "check_attention_strided"]
    rc, lines, err = _oob(mode, names)
    assert rc == 0 and lines == ["OK " + n for n in names], (rc, lines, err)
    names = ["attention_small", "check_attention_strided", "check_layernorm", "check_pre_select", "check_conv3x3", "check_swin_fpn",
             "check_attention_qk_mask", "check_post_golden"]
    rc, lines, err = _oob(mode, names, OPT_IN)
    assert rc == 0 and lines == ["OK " + n for n in names], (rc, lines, err)


def test_guard_pages_do_catch_an_overrun():
    """negative control: mq_layernorm_fwd told about one row more than its input holds dies with SIGSEGV under the guard"""
    import subprocess
    code = (
        "import sys, ctypes, torch\n"
        "sys.path.insert(0, 'tests'); sys.path.insert(0, '.')\n"
        "import simt\nfrom simt import guard\n"
        "lib = simt.library()\n"
        "x = guard.guarded(torch.randn(64, 256).half()); w = guard.guarded(torch.ones(256).half()); y = guard.alloc((65, 256), torch.float16)\n"
        "vp = ctypes.c_void_p\n"
        "rc = lib.mq_layernorm_fwd(vp(x.data_ptr()), 0, vp(0), 0, vp(w.data_ptr()), vp(w.data_ptr()), vp(y.data_ptr()), vp(0), vp(0), {rows}, 256, 1e-5, vp(0))\n"
        "print('returned', rc)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ok = subprocess.run([sys.executable, "-c", code.format(rows=64)], capture_output=True, text=True, cwd=root)
    assert ok.returncode == 0 and "returned 0" in ok.stdout, ok.stderr[-400:]
    bad = subprocess.run([sys.executable, "-c", code.format(rows=65)], capture_output=True, text=True, cwd=root)
    assert bad.returncode == -11, (bad.returncode, bad.stdout, bad.stderr[-300:])


# ---- the FPN output convs as ONE grouped launch of the fused DCNv2 kernel with zero offsets (opt-in: MQ_FPN_VIA_DCN=1)
def test_fpn_convs_through_the_grouped_dcn_kernel(kernels, monkeypatch):
    """a deformable conv sampling at integer positions with corner weights (1, 0, 0, 0) and mask 1 is the plain 3x3 conv, zero padding
    included: same FPN pyramid as through mq_conv3x3_fwd (different fp32 summation order only), stride-2 P6 / P7 too, odd sizes"""
    from mq_det_amd.modeling import pipeline
    spec, sd, cfg, model, P = kernels.tiny(CPU)
    g = torch.Generator().manual_seed(5)
    feats = [torch.randn(2, h, w, c, generator=g).half() for (h, w), c in (((23, 31), 192), ((12, 16), 384), ((6, 8), 768))]
    monkeypatch.setenv("MQ_FPN_VIA_DCN", "0")
    ref = pipeline.fpn_forward(P, feats)
    monkeypatch.setenv("MQ_FPN_VIA_DCN", "1")
    got = pipeline.fpn_forward(P, feats)
    assert [tuple(t.shape) for t in got] == [tuple(t.shape) for t in ref]
    for a, b in zip(got, ref):
        assert float((a.float() - b.float()).abs().max()) <= 2e-3 * max(1.0, float(b.float().abs().max()))
    kernels._CACHE.clear()
    _assert_ok(kernels.check_swin_fpn(CPU))
    kernels._CACHE.clear()
    # round 5: the PLAIN instantiation of the kernel (flags bit 1: one corner per tap gathered, no blend) == the general path, bit for bit,
    # grouped and single launches, stride 1 and 2, patches that straddle the zero padding
    from mq_det_amd import ops
    xs = [torch.randn(2, h, w, 256, generator=g).half() for (h, w) in ((23, 31), (12, 16), (5, 7))]
    wts = [(torch.randn(256, 9 * 256, generator=g) / 48).half() for _ in xs]
    bs = [torch.randn(256, generator=g).half() for _ in xs]
    for stride in (1, 2):
        outs = {}
        for plain in (False, True):
            grp = [dict(x=x, om=pipeline._zero_offsets(2, (x.shape[1] - 1) // stride + 1, (x.shape[2] - 1) // stride + 1, CPU), w=w_, bias=b_,
                        stride=stride, plain=plain) for x, w_, b_ in zip(xs, wts, bs)]
            outs[plain] = [y for (y, _, _) in ops.dcnv2_group(grp, want_stats=False)]
            y1, _ = ops.dcnv2(xs[0], grp[0]["om"], wts[0], bs[0], stride, plain=plain)
            outs[plain].append(y1)
        for a, b in zip(outs[True], outs[False]):
            assert torch.equal(a, b)


# ---- NMS that stops once max_keep boxes of an image are kept (csrc/nms2.hip, opt-in: MQ_NMS_EARLY_STOP=1)
def test_nms_early_stop_keeps_the_same_top_detections(kernels, monkeypatch):
    """score-sorted input: the first max_keep survivors of mq_ml_nms_topk are those of mq_ml_nms, nothing is kept behind the stopping
    chunk, images of one batch stop at different chunks; under permuted wave schedules (the stop flag is read by all five waves of the
    workgroup); then the golden post-processing and the tiny full model with the switch on"""
    import simt
    from mq_det_amd import ops
    g = torch.Generator().manual_seed(13)
    B, N = 3, 1500
    xy = torch.rand(B, N, 2, generator=g) * 300
    boxes = torch.cat([xy, xy + 15 + torch.rand(B, N, 2, generator=g) * 50], -1).contiguous()
    labels = torch.randint(1, 5, (B, N), generator=g, dtype=torch.int32)
    nvalid = torch.tensor([1500, 700, 130], dtype=torch.int32)
    monkeypatch.setenv("MQ_NMS_EARLY_STOP", "0")
    full = ops.ml_nms(boxes, labels, nvalid, 0.6)
    for mode in (("ascending", 0), ("descending", 0), ("random", 7)):
        simt.set_schedule(*mode)
        try:
            for K in (1, 50, 100, 300, 5000):
                monkeypatch.setenv("MQ_NMS_EARLY_STOP", "1")
                part = ops.ml_nms(boxes, labels, nvalid, 0.6, max_keep=K)
                for b in range(B):
                    kf, kp = full[b].nonzero().flatten(), part[b].nonzero().flatten()
                    n = min(K, len(kf))
                    assert len(kp) >= n and torch.equal(kp[:n], kf[:n]), (mode, K, b)          # the K best survivors are the same
                    assert bool((part[b] <= full[b]).all())                                      # and nothing else is ever kept
                    if len(kf) > K:
                        assert len(kp) < len(kf) or K >= len(kf) - 63                            # the sweep really stopped early
        finally:
            simt.set_schedule("ascending")
    monkeypatch.setenv("MQ_NMS_EARLY_STOP", "1")
    _assert_ok(kernels.check_post_golden(CPU))
    _assert_ok(kernels.check_score_agg(CPU))
