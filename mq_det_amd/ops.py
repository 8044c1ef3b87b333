"""ctypes binding of libmqdet_hip.so + thin torch-tensor wrappers (device memory and streams only).

There is NO fallback: if the HIP library is missing or a tensor is not on a GPU these functions raise.
Signatures mirror include/mqdet_hip.h one to one (tests/test_host_cpu.py parses the header and compares).
"""
import ctypes
import math
import os

import torch

_LIB = None
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmqdet_hip.so")

_vp, _i, _l, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
EXPECTED_ABI = 31        # mq_abi_version() of the csrc/ revision the argument lists below were written for (csrc/api.hip)
_SIGNATURES = {
    "mq_abi_version": (_i, []),
    "mq_attn_workspace_bytes": (_l, [_i, _i, _i, _i, _i]),
    "mq_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _l, _vp, _i, _i, _i, _i, _i] + [_l] * 13 + [_f, _f, _i, _vp]),
    "mq_attn_resident_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _l, _i, _i, _i, _i, _i] + [_l] * 13 + [_f, _f, _vp]),
    "mq_patch_embed_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "mq_attn_text_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i] + [_l] * 13 + [_f, _f, _i, _vp]),
    "mq_bert_attn_qkv_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _l, _l, _l, _l, _f, _f, _vp]),
    "mq_attn_chunked_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i] + [_l] * 13 + [_f, _f, _i, _vp]),
    "mq_window_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mq_window_attn_qkv_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mq_gcp_sparse_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mq_gcp_gate_residual_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _l, _i, _i, _vp]),
    "mq_gcp_attn_fwd": (_i, [_vp] * 16 + [_l, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "mq_vlfuse_i2t_fwd": (_i, [_vp, _vp, _vp, _l, _l, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "mq_vlfuse_t2i_workspace_bytes": (_l, [_i, _i, _i]),
    "mq_vlfuse_t2i_fwd": (_i, [_vp, _l, _l, _l, _vp, _vp, _vp, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "mq_layernorm_fwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _l, _i, _f, _vp]),
    "mq_layernorm2_fwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _l, _i, _f, _vp]),
    "mq_layernorm_clamp_fwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _l, _i, _f, _f, _vp]),
    "mq_clamp_gelu_clamp": (_i, [_vp, _vp, _l, _f, _vp]),
    "mq_patch_merge_ln_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "mq_swin_mlp2_fwd": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _l, _i, _i, _vp]),
    "mq_conv3x3_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _i, _i, _i, _vp]),
    "mq_conv3x3_nchw32_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _i, _vp]),
    "mq_conv3x3_nchw32_v2_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _i, _vp]),
    "mq_conv3x3_nchw32_group_fwd": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    "mq_dcnv2_stats_blocks": (_i, [_i, _i, _i]),
    "mq_dcnv2_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _i, _i, _i, _i, _i, _i, _vp]),
    "mq_dcnv2_group_fwd": (_i, [_vp, _i, _vp]),
    "mq_dyconv_stats": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mq_dyconv_coef": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "mq_dyconv_coef_group": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _f, _vp]),
    "mq_dyconv_fuse": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _l, _vp, _i, _i, _i, _i, _vp]),
    "mq_dyrelu_coef": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mq_dyconv_epilogue_group": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mq_add_upsample_nearest": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mq_pool2x2_tokens_fwd": (_i, [_vp, _i, _vp, _i, _i, _vp]),
    "mq_dyrelu_ln_fwd": (_i, [_vp, _l, _vp, _vp, _i, _vp, _vp, _f, _vp, _i, _i, _i, _vp]),
    "mq_dyrelu_apply": (_i, [_vp, _vp, _i, _i, _i, _l, _vp]),
    "mq_align_scores_fwd": (_i, [_vp, _i, _vp, _vp, _l, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _l, _i, _vp]),
    "mq_align_fused_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "mq_box_decode": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _l, _vp]),
    "mq_roi_align_fwd": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _l, _l, _l, _l, _i, _i, _f, _i, _i, _i, _vp]),
    "mq_msdeform_attn_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mq_msdeform_attn_q_fwd": (_i, [_vp, _i, _l, _l, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mq_post_select_workspace_bytes": (_l, [_vp, _vp, _i, _i, _i]),
    "mq_post_select_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mq_post_sort_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mq_post_finalize_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mq_ml_nms_workspace_bytes": (_l, [_i, _i]),
    "mq_ml_nms_topk": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp]),
    "mq_ml_nms": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
}
# entry points with 16-bit operands also exist as <name>_bf16 (same signature; include/mqdet_hip.h MQ_BF16_TWIN)
BF16_TWINS = ("mq_attn_fwd", "mq_attn_resident_fwd", "mq_attn_text_fwd", "mq_bert_attn_qkv_fwd", "mq_patch_embed_fwd", "mq_attn_chunked_fwd", "mq_window_attn_fwd", "mq_window_attn_qkv_fwd", "mq_gcp_sparse_attn_fwd", "mq_gcp_gate_residual_fwd", "mq_gcp_attn_fwd", "mq_vlfuse_i2t_fwd", "mq_vlfuse_t2i_fwd",
              "mq_layernorm_fwd", "mq_layernorm2_fwd", "mq_layernorm_clamp_fwd", "mq_clamp_gelu_clamp", "mq_patch_merge_ln_fwd", "mq_swin_mlp2_fwd", "mq_conv3x3_fwd", "mq_conv3x3_nchw32_fwd", "mq_conv3x3_nchw32_v2_fwd", "mq_conv3x3_nchw32_group_fwd", "mq_dcnv2_fwd", "mq_dcnv2_group_fwd",
              "mq_dyconv_stats", "mq_dyconv_coef", "mq_dyconv_coef_group", "mq_dyconv_fuse", "mq_dyrelu_coef", "mq_dyconv_epilogue_group", "mq_dyrelu_apply", "mq_dyrelu_ln_fwd", "mq_add_upsample_nearest", "mq_pool2x2_tokens_fwd",
              "mq_align_scores_fwd", "mq_align_fused_fwd", "mq_box_decode", "mq_roi_align_fwd", "mq_msdeform_attn_fwd", "mq_msdeform_attn_q_fwd")
# ... and as <name>_f32 (fp32 operands, the precise mode), except ROIAlign: its base entry point already takes fp32 features (is_f32 flag) and has
# no other 16-bit operand.  (Round 6: the MSDeformAttn kernels have the twin -- the fused-query form reads a 16-bit `qproj`, a float there.)
F32_TWINS = tuple(n for n in BF16_TWINS if n not in ("mq_roi_align_fwd",))
for _n in BF16_TWINS:
    _SIGNATURES[_n + "_bf16"] = _SIGNATURES[_n]
for _n in F32_TWINS:
    _SIGNATURES[_n + "_f32"] = _SIGNATURES[_n]
EXPORTS = tuple(_SIGNATURES)


# Kernel selection: defaults, the per-thread live table, configure() / activate() -- mq_det_amd/selection.py (the same objects under the old names)
from .selection import KERNEL_DEFAULTS, KERNELS, _ThreadLocalTable, configure, activate  # noqa: E402,F401


def lib_path():
    return _LIB_PATH


def load_library():
    """dlopen libmqdet_hip.so (built by mq_det_amd.build / __graft_entry__.build).  Raises if absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} not found: the MI355X HIP library is required (python -m mq_det_amd.build); "
                               "there is no CPU / eager fallback for the MQ-Det hot path")
        lib = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)           # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        got = lib.mq_abi_version()
        if got != EXPECTED_ABI:
            # a library from another revision of csrc/ exports the same names with other argument lists: calling it would pass shifted
            # pointers / ints -- refuse instead
            raise RuntimeError(f"{_LIB_PATH} has ABI version {got}, these bindings are written for {EXPECTED_ABI}: "
                               "rebuild it (python -m mq_det_amd.build --force)")
        _LIB = lib
    return _LIB


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- optional per-kernel timing with HIP events on the launch stream (bench.py roofline numbers)
_TIMING = None


def timing_active():
    return _TIMING is not None


def start_timing():
    global _TIMING
    _TIMING = []


def stop_timing():
    """-> {tag: (launches, total_ms, algorithmic_bytes)}; synchronises."""
    global _TIMING
    rec, _TIMING = _TIMING or [], None
    torch.cuda.synchronize()
    out = {}
    for tag, a, b, nb in rec:
        n, t, by = out.get(tag, (0, 0.0, 0))
        out[tag] = (n + 1, t + a.elapsed_time(b), by + nb)
    return out


class _timed:
    """HIP events on the launch stream around one launch; nbytes = the launch's ALGORITHMIC HBM bytes (inputs read once +
    outputs written once) for the HBM-bound kernels' GB/s in bench.py."""

    def __init__(self, tag, nbytes=0):
        self.tag, self.nbytes = tag, nbytes

    def __enter__(self):
        if _TIMING is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if _TIMING is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _TIMING.append((self.tag, self.a, b, int(self.nbytes)))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc}")


class _OperandTypes:
    """The tensor dtypes the kernels take as their MFMA operands: fp16 / bf16, or -- in the precise mode, KERNELS["F32_OPERANDS"] -- fp32."""

    def _cur(self):
        return (torch.float32,) if KERNELS.get("F32_OPERANDS", 0) else (torch.float16, torch.bfloat16)

    def __contains__(self, dtype):
        return dtype in self._cur()

    def __iter__(self):
        return iter(self._cur())

    def __getitem__(self, i):
        return self._cur()[i]

    def __add__(self, other):
        return self._cur() + tuple(other)

    def __radd__(self, other):
        return tuple(other) + self._cur()


_H16 = _OperandTypes()


def f32_operands():
    """0: 16-bit operands; 1: the precise mode on the device (160 KB of LDS per workgroup); 2: the same through the kernel-source emulation."""
    return KERNELS.get("F32_OPERANDS", 0)


def _fn(lib, name, *ts):
    """The entry point for the operand type of `ts`: `name` (fp16), `name_bf16` (the same kernel compiled with bf16 operands,
    include/mqdet_hip.h MQ_BF16_TWIN) or, in the precise mode, `name_f32` (fp32 operands).  All operands of one call must have the same type."""
    if KERNELS.get("F32_OPERANDS", 0):
        # ROIAlign / MSDeformAttn have no *_f32 twin (build.py F32_SKIP) -- their base entry points take fp32 features through the
        # is_f32 flag -- so the precise mode binds the base symbol for them instead of raising AttributeError in extract_query()
        return getattr(lib, name + "_f32") if name in F32_TWINS else getattr(lib, name)
    kinds = {t.dtype for t in ts if t is not None and t.dtype in (torch.float16, torch.bfloat16)}
    if len(kinds) > 1:
        raise TypeError(f"{name}: fp16 and bf16 operands in one call")
    return getattr(lib, name + "_bf16") if torch.bfloat16 in kinds else getattr(lib, name)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mq_det_amd ops need GPU tensors: the hot path has no CPU fallback")


def patch_embed_pack(w, nchw=False):
    """PatchEmbed.proj weight [C, 3, 4, 4] (any float type) -> [C, 64] in the k order of mq_patch_embed_fwd: the 48 inputs of a patch as 12
    chunks of 4 values at k-slots 4 c .. 4 c + 3, k 48 .. 63 zero.  Channels-last pixels (nchw False): chunk c = image row c // 3, values
    4 (c % 3) .. of the row's 12 (px, channel) pairs.  fp32 NCHW pixels (nchw True): chunk c = (channel c // 4, image row c % 4), 4 pixels."""
    C = w.shape[0]
    out = w.new_zeros(C, 64)
    out[:, :48] = w.reshape(C, 48) if nchw else w.permute(0, 2, 3, 1).reshape(C, 48)
    return out.contiguous()


def patch_embed(img, wpk, bias, g0, b0, g1, b1, eps=1e-5):
    """img: [B,Hi,Wi,3] 16-bit contiguous (channels-last pixels, wpk packed with nchw=False) or [B,3,Hi,Wi] fp32 contiguous (wpk packed with
    nchw=True) -> (x32 [B,Hi/4*Wi/4,C] fp32 = LN_0(proj(patches)), h1 16-bit = LN_1(x32)); mq_patch_embed_fwd."""
    lib = load_library()
    _need_gpu(img, wpk, bias, g0, b0, g1, b1)
    f32 = img.dtype == torch.float32
    if f32 and f32_operands() and img.shape[-1] == 3 and img.shape[1] != 3:
        f32 = False           # precise mode: channels-last pixels of the operand type (= float32 there), not the caller's NCHW tensor
    if f32:
        B, Cin, Hi, Wi = img.shape
    else:
        B, Hi, Wi, Cin = img.shape
    C = wpk.shape[0]
    assert Cin == 3 and img.is_contiguous() and (f32 or img.dtype == wpk.dtype) and wpk.dtype in _H16 and wpk.shape == (C, 64) and wpk.is_contiguous()
    assert all(t.dtype == torch.float32 and t.is_contiguous() and t.numel() == C for t in (bias, g0, b0, g1, b1))
    n = (Hi // 4) * (Wi // 4)
    x32 = torch.empty(B, n, C, dtype=torch.float32, device=img.device)
    h1 = torch.empty(B, n, C, dtype=wpk.dtype, device=img.device)
    with _timed(f"patch_embed_c{C}", img.numel() * img.element_size() + x32.numel() * 4 + h1.numel() * 2):
        _chk(_fn(lib, "mq_patch_embed_fwd", wpk)(_ptr(img), int(f32), _ptr(wpk), _ptr(bias), _ptr(g0), _ptr(b0), _ptr(g1), _ptr(b1), _ptr(x32), _ptr(h1),
                                                 B, Hi, Wi, C, float(eps), _stream()), "mq_patch_embed_fwd")
    return x32, h1


def attention_text_fits(T, kv_len=None, max_kv=0):
    """Does mq_attn_text_fwd take this key length?  Always in the 16-bit modes (T <= 256); in the precise mode on the device its K / V tiles are
    80 floats per key: up to 160 live keys fit the LDS (longer captions: q|k GEMM + V^T + mq_attn_resident_fwd)."""
    live = max_kv if (kv_len is not None and 0 < max_kv < T) else T
    return T <= 256 and (f32_operands() != 1 or live <= 160)


FUSED_TEXT_MAX_ROWS = 9216          # text rows (B x T) up to which KERNELS["GCP_ATTN_FUSED"] = 1 takes the fused GCP kernel (B = 64 at 144 rows: the largest measured)
FUSED_BERT_MAX_WORKGROUPS = 768     # (batch item, head) workgroups up to which KERNELS["BERT_ATTN_QKV_FUSED"] = 1 takes the fused BERT kernel (B = 64: the largest measured)


def bert_attention_qkv_fits(T, C, heads, key_bias=None, batch=None):
    """Shapes mq_bert_attn_qkv_fwd takes: BERT-base (C = 768 = 12 x 64), up to 256 tokens (precise mode on the device: up to 160 -- the three
    [T, 80] tiles of a head are 154 KB at fp32), one key bias per (batch item, key).  batch: also apply the size policy of
    KERNELS["BERT_ATTN_QKV_FUSED"] = 1 (the kernel only where it is measured to win: not in the split-precise mode)."""
    ok = (C == 768 and C == 64 * heads and T <= (160 if f32_operands() == 1 else 256) and (key_bias is None or key_bias.dim() == 2))
    if ok and batch is not None and KERNELS["BERT_ATTN_QKV_FUSED"] == 1:
        # split-precise mode: the fused kernel still splits its operands inside mfma16 (335 spilled VGPRs at fp32 fragment sizes); the fp32 library
        # GEMM + mq_attn_text_fwd pair measured +1.4 % end to end (GPU call 4 of round 6) -- the policy takes the pair there (= 2 forces the kernel)
        ok = batch * heads <= FUSED_BERT_MAX_WORKGROUPS and not f32_operands()
    return ok


def bert_attention_qkv(x, wqkv, bqkv, heads, key_bias=None, clamp=0.0, kv_len=None, scale=None, packed=False):
    """BertSelfAttention as one launch (mq_bert_attn_qkv_fwd): x [B,T,C] 16-bit hidden states, wqkv [3C,C] / bqkv [3C] the layer's fused
    q | k | v projection (packed: wqkv already pack_b_fragments(...) -- the pipeline packs once; otherwise re-ordered here, one copy per call),
    key_bias None or [B,T] fp32, kv_len [B] int32 or None -> context [B,T,C] in x's dtype."""
    lib = load_library()
    _need_gpu(x, wqkv, bqkv, key_bias, kv_len)
    B, T, C = x.shape
    assert bert_attention_qkv_fits(T, C, heads, key_bias) and x.dtype in _H16 and x.stride(2) == 1
    wqkv = _b_fragments(wqkv, 3 * C, C, packed)
    assert bqkv.shape == (3 * C,) and bqkv.is_contiguous() and wqkv.dtype == bqkv.dtype == x.dtype
    bias_bs = 0
    if key_bias is not None:
        assert key_bias.dtype == torch.float32 and key_bias.shape == (B, T) and key_bias.stride(1) == 1
        bias_bs = key_bias.stride(0)
    if kv_len is not None:
        assert kv_len.dtype == torch.int32 and kv_len.shape == (B,) and kv_len.is_contiguous()
    o = torch.empty(B, T, C, dtype=x.dtype, device=x.device)
    D = C // heads
    with _timed(f"bert_attn_qkv_t{T}"):
        rc = _fn(lib, "mq_bert_attn_qkv_fwd", x)(_ptr(x), _ptr(wqkv), _ptr(bqkv), _ptr(o), _ptr(key_bias), _ptr(kv_len), B, T, C, heads,
                                                 x.stride(0), x.stride(1), o.stride(0), o.stride(1), bias_bs,
                                                 float(scale if scale is not None else 1.0 / math.sqrt(D)), float(clamp), _stream())
    _chk(rc, "mq_bert_attn_qkv_fwd")
    return o


def attention_text(qkv, heads, key_bias=None, clamp=0.0, kv_len=None, max_kv=0, scale=None):
    """Self-attention of the text tokens straight from the fused projection qkv [B,T,3*H*D] (q | k | v along the last dimension), V
    row-major (mq_attn_text_fwd).  key_bias None, [B,T] or [B,H,T] fp32; kv_len [B] int32; max_kv: host bound on kv_len (0 = T).
    -> [B,T,H*D] in qkv's dtype."""
    lib = load_library()
    _need_gpu(qkv, key_bias, kv_len)
    B, T, C3 = qkv.shape
    HD = C3 // 3
    D = HD // heads
    assert C3 == 3 * heads * D and qkv.dtype in _H16 and qkv.stride(2) == 1 and T <= 256 and D in (32, 64)
    assert attention_text_fits(T, kv_len, max_kv), "precise mode: mq_attn_text_fwd_f32 holds up to 160 live keys in LDS"
    if kv_len is not None:
        assert kv_len.dtype == torch.int32 and kv_len.shape == (B,) and kv_len.is_contiguous()
        # max_kv sizes the LDS tiles and the key-block loop: it must cover every kv_len[b] (a smaller bound silently drops live keys -- ADVICE r4).
        # Checked on the host only in debug runs (MQ_DEBUG_SYNC=1): the comparison reads a device tensor
        if max_kv > 0 and os.environ.get("MQ_DEBUG_SYNC") == "1":
            assert int(kv_len.max()) <= max(max_kv, 1), f"attention_text: max_kv = {max_kv} < kv_len.max() = {int(kv_len.max())}"
    bias_bs = bias_hs = 0
    if key_bias is not None:
        assert key_bias.dtype == torch.float32 and key_bias.shape[-1] == T and key_bias.stride(-1) == 1
        if key_bias.dim() == 2:
            bias_bs = key_bias.stride(0)
        else:
            bias_bs, bias_hs = key_bias.stride(0), key_bias.stride(1)
    o = torch.empty(B, T, HD, dtype=qkv.dtype, device=qkv.device)
    q, k, v = qkv[:, :, :HD], qkv[:, :, HD:2 * HD], qkv[:, :, 2 * HD:]
    bs, rs = qkv.stride(0), qkv.stride(1)
    with _timed(f"attn_text_d{D}_nq{T}_nk{T}"):
        rc = _fn(lib, "mq_attn_text_fwd", qkv)(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(key_bias), _ptr(kv_len), B, heads, T, T, D,
                                               bs, rs, D, bs, rs, D, bs, rs, D, o.stride(0), o.stride(1), bias_bs, bias_hs,
                                               float(scale if scale is not None else 1.0 / math.sqrt(D)), float(clamp), int(max_kv), _stream())
    _chk(rc, "mq_attn_text_fwd")
    return o


def attention4(q4, k4, vt4, key_bias=None, scale=None, clamp=0.0, nsplit=1, nk=None, kv_len=None, qk_mask=None):
    """General strided form.  q4 [B,Nq,H,D], k4 [B,Nk,H,D], vt4 [B,H,D,Nk_pad] fp16 views with unit last stride
    (a head stride of 0, e.g. from .expand(), shares the operand across heads); key_bias None, [B,Nk] or [B,H,Nk]
    fp32; qk_mask None or uint8 / bool [B,H,Nq,Nk] view (expand()-ed batch / head dims allowed), 1 = key hidden from query.
    Returns [B,Nq,H*D] fp16."""
    lib = load_library()
    _need_gpu(q4, k4, vt4, key_bias, kv_len)
    B, Nq, H, D = q4.shape
    if kv_len is not None:
        assert kv_len.dtype == torch.int32 and kv_len.shape == (B,) and kv_len.is_contiguous()
    Nk = k4.shape[1] if nk is None else nk
    assert k4.shape[0] == B and k4.shape[2:] == (H, D) and vt4.shape[:3] == (B, H, D) and vt4.shape[3] >= Nk
    assert q4.dtype == k4.dtype == vt4.dtype and q4.dtype in _H16
    assert q4.stride(3) == 1 and k4.stride(3) == 1 and vt4.stride(3) == 1
    bias_bs = bias_hs = 0
    if key_bias is not None:
        assert key_bias.dtype == torch.float32 and key_bias.shape[-1] == Nk and key_bias.stride(-1) == 1
        if key_bias.dim() == 2:
            assert key_bias.shape == (B, Nk)
            bias_bs = key_bias.stride(0)
        else:
            assert key_bias.shape == (B, H, Nk)
            bias_bs, bias_hs = key_bias.stride(0), key_bias.stride(1)
    mask_bs = mask_hs = mask_rs = 0
    if qk_mask is not None:
        _need_gpu(qk_mask)
        if qk_mask.dtype == torch.bool:
            qk_mask = qk_mask.view(torch.uint8)
        assert qk_mask.dtype == torch.uint8 and qk_mask.shape == (B, H, Nq, Nk) and qk_mask.stride(3) == 1
        mask_bs, mask_hs, mask_rs = qk_mask.stride(0), qk_mask.stride(1), qk_mask.stride(2)
    o = torch.empty(B, Nq, H * D, dtype=q4.dtype, device=q4.device)
    mask_ok = qk_mask is None or (Nk % 4 == 0 and mask_bs % 4 == 0 and mask_hs % 4 == 0 and mask_rs % 4 == 0 and qk_mask.data_ptr() % 4 == 0)
    if nsplit <= 1 and Nk <= 256 and D in (32, 64) and mask_ok and o.stride(1) % 4 == 0 and KERNELS["ATTN_RESIDENT"] == 1:
        # text-sized attention on the resident-key kernel (csrc/attn_resident.hip): 1.9x faster on its launches than mq_attn_fwd
        with _timed(f"attn_res_d{D}_nq{Nq}_nk{Nk}"):
            rc = _fn(lib, "mq_attn_resident_fwd", q4)(
                _ptr(q4), _ptr(k4), _ptr(vt4), _ptr(o), _ptr(key_bias), _ptr(kv_len), _ptr(qk_mask), mask_bs, mask_hs, mask_rs,
                B, H, Nq, Nk, D, q4.stride(0), q4.stride(1), q4.stride(2), k4.stride(0), k4.stride(1), k4.stride(2),
                vt4.stride(0), vt4.stride(2), vt4.stride(1), o.stride(0), o.stride(1), bias_bs, bias_hs,
                float(scale if scale is not None else 1.0 / math.sqrt(D)), float(clamp), _stream())
        _chk(rc, "mq_attn_resident_fwd")
        return o
    ws = None
    if nsplit > 1:
        ws = torch.empty(lib.mq_attn_workspace_bytes(B, H, Nq, D, nsplit) // 4, dtype=torch.float32, device=q4.device)
    if qk_mask is None and D in (32, 64) and o.stride(1) % 4 == 0 and KERNELS["ATTN_RESIDENT"] == 1:
        # long key sequences / key splits on the chunked S^T kernel (csrc/attn_resident.hip)
        with _timed(f"attn_chk_d{D}_nq{Nq}_nk{Nk}_s{nsplit}"):
            rc = _fn(lib, "mq_attn_chunked_fwd", q4)(
                _ptr(q4), _ptr(k4), _ptr(vt4), _ptr(o), _ptr(key_bias), _ptr(kv_len), _ptr(ws), B, H, Nq, Nk, D,
                q4.stride(0), q4.stride(1), q4.stride(2), k4.stride(0), k4.stride(1), k4.stride(2),
                vt4.stride(0), vt4.stride(2), vt4.stride(1), o.stride(0), o.stride(1), bias_bs, bias_hs,
                float(scale if scale is not None else 1.0 / math.sqrt(D)), float(clamp), int(nsplit), _stream())
        _chk(rc, "mq_attn_chunked_fwd")
        return o
    with _timed(f"attn_d{D}_nq{Nq}_nk{Nk}_s{nsplit}"):
        rc = _fn(lib, "mq_attn_fwd", q4)(_ptr(q4), _ptr(k4), _ptr(vt4), _ptr(o), _ptr(key_bias), _ptr(kv_len), _ptr(qk_mask), mask_bs, mask_hs,
                             mask_rs, _ptr(ws), B, H, Nq, Nk, D,
                             q4.stride(0), q4.stride(1), q4.stride(2), k4.stride(0), k4.stride(1), k4.stride(2),
                             vt4.stride(0), vt4.stride(2), vt4.stride(1), o.stride(0), o.stride(1), bias_bs, bias_hs,
                             float(scale if scale is not None else 1.0 / math.sqrt(D)), float(clamp), int(nsplit), _stream())
    _chk(rc, "mq_attn_fwd")
    return o


def attention(q, k, vt, num_heads, head_dim, key_bias=None, scale=None, clamp=0.0, nsplit=1, nk=None, kv_len=None, qk_mask=None):
    """q [B,Nq,H*D], k [B,Nk,H*D], vt [B,H*D,Nk_pad] (V transposed, Nk_pad % 8 == 0) fp16 -> [B,Nq,H*D] fp16."""
    B, Nq, HD = q.shape
    H, D = num_heads, head_dim
    assert HD == H * D and k.shape[2] == HD and vt.shape[1] == HD
    assert q.stride(2) == 1 and k.stride(2) == 1 and vt.stride(2) == 1
    q4 = q.as_strided((B, Nq, H, D), (q.stride(0), q.stride(1), D, 1), q.storage_offset())
    k4 = k.as_strided((B, k.shape[1], H, D), (k.stride(0), k.stride(1), D, 1), k.storage_offset())
    vt4 = vt.as_strided((B, H, D, vt.shape[2]), (vt.stride(0), D * vt.stride(1), vt.stride(1), 1), vt.storage_offset())
    return attention4(q4, k4, vt4, key_bias, scale, clamp, nsplit, nk, kv_len, qk_mask)


def window_pad(ws):
    """Padded window length the kernel works on: 64 for N = ws*ws <= 64 (window 7), 160 for N <= 160 (Swin-L, window 12)."""
    n = ws * ws
    if n > 160:
        raise ValueError(f"window {ws}x{ws} = {n} tokens: mq_window_attn_fwd supports up to 160")
    return 64 if n <= 64 else 160


def pad_rel_bias(rel_bias, ws=None):
    """[heads, N, N] relative-position bias -> the zero-padded [heads, NP, NP] table mq_window_attn_fwd reads with 16-byte loads."""
    h, n, _ = rel_bias.shape
    npad = window_pad(ws) if ws is not None else (64 if n <= 64 else 160)
    out = rel_bias.new_zeros(h, npad, npad)
    out[:, :n, :n] = rel_bias
    return out.contiguous()


def window_attention(qkv, qkv_bias, rel_bias, heads, ws, shift):
    """qkv [B,H,W,3C] fp16, qkv_bias [3C] fp16, rel_bias [heads,NP,NP] fp32 (rows = query, cols = key, zero-padded from
    N = ws*ws to NP = window_pad(ws); a [heads,N,N] table is padded here) -> [B,H,W,C] fp16."""
    lib = load_library()
    _need_gpu(qkv, qkv_bias, rel_bias)
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    NP = window_pad(ws)
    assert qkv.is_contiguous() and qkv.dtype in _H16 and qkv_bias.dtype == qkv.dtype
    if rel_bias.shape == (heads, ws * ws, ws * ws) and ws * ws != NP:
        rel_bias = pad_rel_bias(rel_bias, ws)
    assert rel_bias.dtype == torch.float32 and rel_bias.is_contiguous() and rel_bias.shape == (heads, NP, NP)
    out = torch.empty(B, H, W, C, dtype=qkv.dtype, device=qkv.device)
    with _timed(f"window_attn_c{C}", qkv.numel() * 2 + out.numel() * 2):
        _chk(_fn(lib, "mq_window_attn_fwd", qkv)(_ptr(qkv), _ptr(qkv_bias), _ptr(rel_bias), _ptr(out), B, H, W, C, heads, ws, shift,
                                    _stream()), "mq_window_attn_fwd")
    return out


WINDOW_QKV_WIDTHS = (96, 192)      # widths mq_window_attn_qkv_fwd is instantiated for (192: weights streamed per head)


def window_qkv_fused(C, ws, numel=0):
    """Does the kernel selection route a Swin block of width C / window ws through mq_window_attn_qkv_fwd?  SWIN_QKV_FUSED: 0 never,
    1 the resident-weight width (96), 2 also the streamed one (192).  numel: elements of the [B,H,W,C] input (the kernel addresses it
    with 32-bit offsets)."""
    k = KERNELS["SWIN_QKV_FUSED"]
    return ws * ws <= 64 and numel < 2 ** 31 and ((k >= 1 and C == 96) or (k >= 2 and C == 192))



def window_attention_qkv(x, w, bias, rel_bias, heads, ws, shift):
    """Window attention with the qkv projection inside (mq_window_attn_qkv_fwd): x [B,H,W,C] 16-bit = norm1(x), w [3C,C] / bias [3C] =
    attn.qkv, rel_bias as window_attention -> [B,H,W,C].  C in WINDOW_QKV_WIDTHS, ws * ws <= 64."""
    lib = load_library()
    _need_gpu(x, w, bias, rel_bias)
    B, H, W, C = x.shape
    NP = window_pad(ws)
    assert C in WINDOW_QKV_WIDTHS and NP == 64 and heads * 32 == C
    assert x.is_contiguous() and x.dtype in _H16 and w.dtype == x.dtype == bias.dtype and w.is_contiguous() and w.shape == (3 * C, C)
    if rel_bias.shape == (heads, ws * ws, ws * ws) and ws * ws != NP:
        rel_bias = pad_rel_bias(rel_bias, ws)
    assert rel_bias.dtype == torch.float32 and rel_bias.is_contiguous() and rel_bias.shape == (heads, NP, NP)
    out = torch.empty(B, H, W, C, dtype=x.dtype, device=x.device)
    with _timed(f"window_attn_qkv_c{C}", 2 * x.numel() * 2):
        _chk(_fn(lib, "mq_window_attn_qkv_fwd", x)(_ptr(x), _ptr(w), _ptr(bias), _ptr(rel_bias), _ptr(out), B, H, W, C, heads, ws, shift,
                                                   _stream()), "mq_window_attn_qkv_fwd")
    return out


def gcp_sparse_attention(q, kv, idx, heads=8, dim_head=64):
    """q [B,T,512] fp16, kv [B,V,1024] fp16, idx [B,T,S] int32 (-1 pad) -> [B,T,512] fp16."""
    lib = load_library()
    _need_gpu(q, kv, idx)
    B, T, HD = q.shape
    V, S = kv.shape[1], idx.shape[2]
    assert q.is_contiguous() and kv.is_contiguous() and idx.is_contiguous() and idx.dtype == torch.int32
    assert q.dtype == kv.dtype and q.dtype in _H16 and kv.shape[2] == 2 * HD
    out = torch.empty_like(q)
    with _timed(f"gcp_sparse_attn_s{S}", q.numel() * 4 + kv.numel() * 2):
        _chk(_fn(lib, "mq_gcp_sparse_attn_fwd", q)(_ptr(q), _ptr(kv), _ptr(idx), _ptr(out), B, T, V, S, heads, dim_head, _stream()),
             "mq_gcp_sparse_attn_fwd")
    return out


def gcp_attention_fits(x, idx, policy=False):
    """Shapes mq_gcp_attn_fwd takes: the fp32 text stream of BERT-base width, at most 8 vision-query slots per token.  policy: also apply the
    size rule of KERNELS["GCP_ATTN_FUSED"] = 1 (up to FUSED_TEXT_MAX_ROWS text rows per launch)."""
    ok = x.dtype == torch.float32 and x.shape[-1] == 768 and idx.shape[-1] <= 8
    if ok and policy and KERNELS["GCP_ATTN_FUSED"] == 1:
        # (split-precise mode: the eight launches, +1.3 % end to end -- as for the fused BERT kernel above)
        ok = x.numel() // 768 <= FUSED_TEXT_MAX_ROWS and not f32_operands()
    return ok


def pack_b_fragments(w):
    """W [N, K] (an nn.Linear weight) -> the same elements in MFMA B-fragment order [N/16, K/32, 64, 8] (include/mqdet_hip.h): what the fused text
    kernels stream.  Done once per weight when the model is packed (modeling/pipeline.py); any dtype."""
    N, K = w.shape
    assert N % 16 == 0 and K % 32 == 0
    return w.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(N // 16, K // 32, 64, 8)


def unpack_b_fragments(wp):
    """The inverse of pack_b_fragments: [N/16, K/32, 64, 8] -> the row-major [N, K] weight (tests, the torch stand-ins of the kernels)."""
    nt, ks = wp.shape[:2]
    return wp.view(nt, ks, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(nt * 16, ks * 32)


def _b_fragments(w, N, K, packed):
    if packed:
        assert w.shape == (N // 16, K // 32, 64, 8) and w.is_contiguous(), "expected pack_b_fragments(weight)"
        return w
    assert w.shape == (N, K)
    return pack_b_fragments(w)


def gcp_attention(x, kv, idx, wq, wout, wg1, w2, ln_a, ln_g, ln_f=None, eps=1e-5, want_gate=False, rows_per_block=0, packed=False):
    """The attention half of a GatedCrossAttentionBlock in one launch (mq_gcp_attn_fwd).  x [B,T,768] fp32 residual stream, kv [B,V,1024] 16-bit,
    idx [B,T,S] int32, wq [512,768], wout [768,512], wg1 [384,768] (packed: each already pack_b_fragments(...) -- the pipeline packs once; otherwise
    re-ordered here, three small copies per call), w2 [384]; ln_a / ln_g / ln_f = (gamma, beta) of the attention-input, gate-input
    and (optional) feed-forward-input LayerNorms -> x_out [B,T,768] fp32 (, y = LN_f(x_out) 16-bit when ln_f) (, gate [B,T] fp32 when want_gate)."""
    lib = load_library()
    _need_gpu(x, kv, idx, wq, wout, wg1, w2)
    B, T, C = x.shape
    V, S = kv.shape[1], idx.shape[2]
    assert gcp_attention_fits(x, idx) and x.is_contiguous() and kv.shape == (B, V, 1024) and kv.is_contiguous() and kv.dtype in _H16
    assert idx.dtype == torch.int32 and idx.shape[:2] == (B, T) and idx.is_contiguous()
    wq, wout, wg1 = _b_fragments(wq, 512, C, packed), _b_fragments(wout, C, 512, packed), _b_fragments(wg1, 384, C, packed)
    assert w2.numel() == 384
    ws = [wq, wout, wg1, w2, *ln_a, *ln_g] + (list(ln_f) if ln_f is not None else [])
    assert all(w.dtype == kv.dtype and w.is_contiguous() for w in ws)
    out = torch.empty_like(x)
    y = torch.empty(B, T, C, dtype=kv.dtype, device=x.device) if ln_f is not None else None
    gate = torch.empty(B, T, dtype=torch.float32, device=x.device) if want_gate else None
    with _timed(f"gcp_attn_fused_s{S}"):
        rc = _fn(lib, "mq_gcp_attn_fwd", kv)(_ptr(x), _ptr(out), _ptr(y), _ptr(gate), _ptr(kv), _ptr(idx), _ptr(wq), _ptr(wout), _ptr(wg1), _ptr(w2),
                                            _ptr(ln_a[0]), _ptr(ln_a[1]), _ptr(ln_g[0]), _ptr(ln_g[1]),
                                            _ptr(ln_f[0]) if ln_f is not None else _vp(0), _ptr(ln_f[1]) if ln_f is not None else _vp(0),
                                            B * T, T, V, S, C, 8, 64, 384, float(eps), int(rows_per_block), _stream())
    _chk(rc, "mq_gcp_attn_fwd")
    res = (out,) + ((y,) if y is not None else ()) + ((gate,) if want_gate else ())
    return res if len(res) > 1 else out


def gcp_gate_residual(sup, h, w2, x, want_gate=False):
    """out = sup * tanh(w2 . gelu(h)) + x ; sup [..., C], h [..., G], w2 [G] fp16; x (the residual stream) fp16 or fp32,
    the result has x's dtype."""
    lib = load_library()
    _need_gpu(sup, h, w2, x)
    C, G = sup.shape[-1], h.shape[-1]
    M = sup.numel() // C
    assert sup.is_contiguous() and h.is_contiguous() and x.is_contiguous() and w2.is_contiguous()
    assert sup.dtype == h.dtype == w2.dtype and sup.dtype in _H16 and x.dtype in (sup.dtype, torch.float32)
    out = torch.empty_like(x)
    gate = torch.empty(M, dtype=torch.float32, device=x.device) if want_gate else None
    _chk(_fn(lib, "mq_gcp_gate_residual_fwd", sup)(_ptr(sup), _ptr(h), _ptr(w2), _ptr(x), int(x.dtype == torch.float32), _ptr(out), _ptr(gate),
                                      M, C, G, _stream()), "mq_gcp_gate_residual_fwd")
    return (out, gate) if want_gate else out


def _kv_strides(kf, vo=None):
    """Element strides (batch, head, token) of the folded keys / values: contiguous [B,heads,T,256] tensors or equally strided views of one
    projection output (pipeline.vl_text_prep); rows of 256 consecutive elements, everything a multiple of 16 bytes."""
    st = kf.stride()
    unit = 16 // kf.element_size()
    assert st[3] == 1 and st[2] >= 256 and all(x % unit == 0 for x in st[:3]) and kf.data_ptr() % 16 == 0, st
    if vo is not None:
        assert vo.stride() == st and vo.data_ptr() % 16 == 0, (vo.stride(), st)
    return int(st[0]), int(st[1]), int(st[2])


def vlfuse_i2t(v_ln, kf, vo, bias, out_bias, kv_len=None, max_kv=0, clamp=50000.0, variant=None):
    """VLFuse image side (mq_vlfuse_i2t_fwd).  v_ln [B,N,256], kf / vo [B,heads,T,256] fp16 (heads <= 8; contiguous or equally strided views, _kv_strides), bias [B,heads,T] fp32 or None,
    out_bias [256] fp16, kv_len [B] int32 or None (max_kv: host-side upper bound, 0 = T) -> [B,N,256] fp16:
    v_ln + out_bias + sum_h softmax_t(clamp(v_ln.kf_h + bias_h)) vo_h."""
    lib = load_library()
    _need_gpu(v_ln, kf, vo, out_bias)
    B, N, C = v_ln.shape
    Hh, T = kf.shape[1], kf.shape[2]
    assert C == 256 and kf.shape == (B, Hh, T, 256) and vo.shape == kf.shape and T <= 256 and 1 <= Hh <= 8
    assert v_ln.is_contiguous() and out_bias.is_contiguous()
    kv_bs, kv_hs, kv_ts = _kv_strides(kf, vo)
    assert v_ln.dtype == kf.dtype == vo.dtype == out_bias.dtype and v_ln.dtype in _H16
    if bias is not None:
        assert bias.shape == (B, Hh, T) and bias.dtype == torch.float32 and bias.is_contiguous()
    if kv_len is not None:
        assert kv_len.dtype == torch.int32 and kv_len.numel() == B and kv_len.is_contiguous()
    live = max_kv if (kv_len is not None and 0 < max_kv < T) else T
    # (round 5's precise mode sent captions of more than 160 tokens through a plain torch statement of this sum: its fp32 Q tile did not fit the
    # LDS beside two K / V tiles.  The split-precise kernels keep Q in registers for every caption length: one path.)
    out = torch.empty_like(v_ln)
    variant = KERNELS["VLFUSE_I2T_VARIANT"] if variant is None else int(variant)
    if f32_operands():
        variant = 0           # split-precise mode: Q fragments always in registers
    with _timed(f"vlfuse_i2t_n{N}_t{T}"):
        _chk(_fn(lib, "mq_vlfuse_i2t_fwd", v_ln)(_ptr(v_ln), _ptr(kf), _ptr(vo), kv_bs, kv_hs, kv_ts, _ptr(bias), _ptr(kv_len), _ptr(out_bias), _ptr(out),
                                   B, N, T, Hh, int(max_kv), float(clamp), int(variant),
                                   _stream()), "mq_vlfuse_i2t_fwd")
    return out


def vlfuse_t2i(kf, v_ln, nsplit, clamp=50000.0, kv_len=None, key_mask=None, max_kv=0, variant=0):
    """VLFuse text side (mq_vlfuse_t2i_fwd).  kf [B,heads,T,256] (queries), v_ln [B,N,256] (keys = values) fp16
    -> [B,T,heads*256] fp16 = softmax_n(clamp(kf.v_ln)) v_ln per head.  kv_len [B] int32: 16-row blocks of pure padding
    (rows >= kv_len[b]) are skipped and returned as zeros; max_kv: host-side bound of kv_len (0 = T), sizes the grid.  key_mask uint8 [B, >= 64*ceil(N/64)] (row stride % 4 == 0):
    1 = image token is padding (see image_key_mask)."""
    lib = load_library()
    _need_gpu(kf, v_ln, key_mask)
    B, N, C = v_ln.shape
    Hh, T = kf.shape[1], kf.shape[2]
    assert C == 256 and kf.shape == (B, Hh, T, 256) and v_ln.is_contiguous() and 1 <= Hh <= 8
    kv_bs, kv_hs, kv_ts = _kv_strides(kf)
    km_bs = 0
    if key_mask is not None:
        assert key_mask.dtype == torch.uint8 and key_mask.dim() == 2 and key_mask.shape[0] == B and key_mask.stride(1) == 1
        km_bs = key_mask.stride(0)
        assert km_bs % 4 == 0 and key_mask.shape[1] >= -(-N // 64) * 64
    assert kf.dtype == v_ln.dtype and kf.dtype in _H16
    nsplit = max(1, int(nsplit))
    if kv_len is not None:
        assert kv_len.dtype == torch.int32 and kv_len.numel() == B and kv_len.is_contiguous()
    else:
        max_kv = 0
    ws = torch.empty(lib.mq_vlfuse_t2i_workspace_bytes(B, T, nsplit) // 4, dtype=torch.float32, device=kf.device)
    out = torch.empty(B, T, Hh * 256, dtype=kf.dtype, device=kf.device)
    with _timed(f"vlfuse_t2i_n{N}_t{T}_s{nsplit}"):
        _chk(_fn(lib, "mq_vlfuse_t2i_fwd", kf)(_ptr(kf), kv_bs, kv_hs, kv_ts, _ptr(v_ln), _ptr(kv_len), _ptr(key_mask), km_bs, _ptr(ws), _ptr(out), B, N, T, Hh,
                                   nsplit, int(max_kv), float(clamp), int(variant), _stream()), "mq_vlfuse_t2i_fwd")
    return out


def image_key_mask(mask):
    """bool [B, N] (True = padding token) -> the uint8 [B, 64*ceil(N/64)] layout mq_vlfuse_t2i_fwd reads with 4-byte loads."""
    B, N = mask.shape
    out = torch.ones(B, -(-N // 64) * 64, dtype=torch.uint8, device=mask.device)
    out[:, :N] = mask.to(torch.uint8)
    return out


def layer_norm(x, gamma, beta, eps=1e-5, residual=None, want_sum=True, want_y32=False, want_y=True, clamp=0.0):
    """LayerNorm over the last dim (mq_layernorm_fwd).  x: contiguous fp16 or fp32; residual (optional, same shape):
    fp16 or fp32, s = x + residual is normalised.  Returns, in this order and only those asked for:
      y   fp16  (want_y)              -- GEMM operand
      y32 fp32  (want_y32)            -- unrounded, the post-LN residual stream
      sum       (residual given and want_sum) -- s: fp32 if x or residual is fp32, else fp16 (s rounded before the statistics)
    clamp > 0 (mq_layernorm_clamp_fwd): x is clamped to +-clamp before the residual add, y / y32 after the affine.
    A single result is returned bare, several as a tuple."""
    lib = load_library()
    _need_gpu(x, gamma, beta, residual)
    C = x.shape[-1]
    h16 = gamma.dtype                                    # the 16-bit type of this call: fp16 or bf16 (weights decide)
    assert x.is_contiguous() and x.dtype in (h16, torch.float32)
    assert h16 in _H16 and beta.dtype == h16
    rows = x.numel() // C
    xf = x.dtype == torch.float32
    rf = False
    y = torch.empty(x.shape, dtype=h16, device=x.device) if want_y else None
    y32 = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_y32 else None
    xsum = None
    if residual is not None:
        assert residual.shape == x.shape and residual.is_contiguous() and residual.dtype in (h16, torch.float32)
        rf = residual.dtype == torch.float32
        if want_sum:
            xsum = torch.empty(x.shape, dtype=torch.float32 if (xf or rf) else h16, device=x.device)
    with _timed(f"layernorm_c{C}", sum(t.numel() * t.element_size() for t in (x, residual, y, y32, xsum) if t is not None)):
        # LN_VARIANT 2: the load-batched kernel of csrc/layernorm2.hip (same results bit for bit as mq_layernorm_fwd)
        name = "mq_layernorm2_fwd" if KERNELS["LN_VARIANT"] == 2 else "mq_layernorm_fwd"
        if clamp > 0:
            _chk(_fn(lib, "mq_layernorm_clamp_fwd", gamma)(_ptr(x), int(xf), _ptr(residual), int(rf), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(y32), _ptr(xsum),
                                                           rows, C, float(eps), float(clamp), _stream()), "mq_layernorm_clamp_fwd")
        else:
            _chk(_fn(lib, name, gamma)(_ptr(x), int(xf), _ptr(residual), int(rf), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(y32), _ptr(xsum),
                                       rows, C, float(eps), _stream()), name)
    out = [t for t in (y, y32, xsum) if t is not None]
    return out[0] if len(out) == 1 else tuple(out)


def clamp_gelu_clamp(x, clamp):
    """clamp(gelu(clamp(x))) elementwise on a contiguous 16-bit tensor (mq_clamp_gelu_clamp): exact GELU in fp32, rounded once."""
    lib = load_library()
    _need_gpu(x)
    assert x.is_contiguous() and x.dtype in _H16 and x.numel() % 8 == 0 and clamp > 0
    out = torch.empty_like(x)
    with _timed("clamp_gelu_clamp", 2 * x.numel() * 2):
        _chk(_fn(lib, "mq_clamp_gelu_clamp", x)(_ptr(x), _ptr(out), x.numel(), float(clamp), _stream()), "mq_clamp_gelu_clamp")
    return out


def patch_merge_ln(x, gamma, beta, eps=1e-5):
    """Swin PatchMerging gather + LayerNorm in one kernel (mq_patch_merge_ln_fwd): x [B,H,W,C] fp16 / bf16 / fp32 contiguous,
    gamma / beta [4C] -> y [B, ceil(H/2)*ceil(W/2), 4C] in gamma's dtype (= F.pad + four strided slices + cat + LayerNorm)."""
    lib = load_library()
    _need_gpu(x, gamma, beta)
    B, H, W, C = x.shape
    h16 = gamma.dtype
    assert x.is_contiguous() and x.dtype in (h16, torch.float32) and h16 in _H16 and beta.dtype == h16
    assert gamma.numel() == 4 * C and beta.numel() == 4 * C and C % 8 == 0 and 4 * C <= 3072
    y = torch.empty(B, ((H + 1) // 2) * ((W + 1) // 2), 4 * C, dtype=h16, device=x.device)
    with _timed(f"patch_merge_ln_c{C}", x.numel() * x.element_size() + y.numel() * 2):
        _chk(_fn(lib, "mq_patch_merge_ln_fwd", gamma)(_ptr(x), int(x.dtype == torch.float32), _ptr(gamma), _ptr(beta), _ptr(y), B, H, W, C,
                                                     float(eps), _stream()), "mq_patch_merge_ln_fwd")
    return y


SWIN_MLP_WIDTHS = (96, 192, 384)


def swin_mlp_w2_perm(K, device=None):
    """Index tensor of the k-slot permutation of fc2.weight inside a 32-block (the order the GELU epilogue leaves the hidden units in the
    MFMA accumulator): w2p = w2[:, perm]; swin_mlp2_pack applies it."""
    k = torch.arange(K, device=device)
    blk, slot = k >> 5, k & 31
    g, t = slot >> 3, slot & 7
    return blk * 32 + torch.where(t < 4, 4 * g + t, 16 + 4 * g + (t - 4))


def split_planar_blocks(t, block=512):
    """The split-precise form of an fp32 operand tensor whose kernel reads it in blocks of `block` elements (csrc/common.h: x = hi + lo / 2^11,
    hi = fp16(x), lo = fp16((x - hi) 2^11)): same shape, dtype float32, but the BYTES of every block are [hi: block fp16 | lo: block fp16] -- what a
    kernel of the precise mode stages with two linear copies and reads back as one 16-byte fragment per plane.  Done once, at pack time."""
    assert t.dtype == torch.float32 and t.numel() % block == 0
    x = t.reshape(-1, block)
    hi = x.to(torch.float16)
    lo = ((x - hi.float()) * 2048.0).to(torch.float16)
    return torch.cat([hi, lo], 1).contiguous().view(torch.float32).reshape(t.shape)


def unsplit_planar_blocks(t, block=512):
    """Inverse of split_planar_blocks (tests): the fp32 values hi + lo / 2^11."""
    h = t.reshape(-1, block).contiguous().view(torch.float16).reshape(-1, 2, block).float()
    return (h[:, 0] + h[:, 1] / 2048.0).reshape(t.shape)


def swin_mlp2_pack(w1, w2):
    """fc1.weight [4C, C], fc2.weight [C, 4C] -> (w1f, w2f), the fragment-major operands of mq_swin_mlp2_fwd (include/mqdet_hip.h):
    every 512-element block is one MFMA A fragment in lane order (lane = 16 g + l15 holds row l15, k-slots 8 g .. 8 g + 7); w1f carries
    two zero chunks behind the last one, w2f the k-slot permutation of swin_mlp_w2_perm."""
    HID, C = w1.shape
    assert w2.shape == (C, HID) and C % 32 == 0 and HID % 32 == 0
    KS, CT, NCH = C // 32, C // 16, HID // 32
    w1f = w1.reshape(NCH, 2, 16, KS, 4, 8).permute(0, 1, 3, 4, 2, 5).reshape(NCH, -1)           # [j][hb][ks][g][l15][8]
    w1f = torch.cat([w1f, w1f.new_zeros(2, w1f.shape[1])], 0).reshape(-1).contiguous()
    w2p = w2[:, swin_mlp_w2_perm(HID, w2.device)]
    w2f = w2p.reshape(CT, 16, NCH, 4, 8).permute(2, 0, 3, 1, 4).reshape(-1).contiguous()         # [j][ct][g][l15][8]
    if w1f.dtype == torch.float32 and f32_operands():
        # split-precise mode: every weight fragment feeds ONE MFMA per wave, so the operands are split HERE, once (csrc/swin_mlp2.hip MQ_SW_SPLIT)
        w1f, w2f = split_planar_blocks(w1f), split_planar_blocks(w2f)
    return w1f, w2f


def swin_mlp2(x, delta, ln_g, ln_b, eps, w1f, b1, w2f, b2, next_ln=None, flags=None, into=None):
    """Fused Swin MLP half, second generation (mq_swin_mlp2_fwd): arguments as swin_mlp but (w1f, w2f) = swin_mlp2_pack(fc1.weight,
    fc2.weight); flags (default KERNELS["SWIN_MLP2_FLAGS"]): bit 1 = table GELU in the main kernel, bit 0 = no pass / tail split, bit 2 =
    every block through the tail kernel, bit 3 / bit 4 = only the main-kernel / only the tail blocks (`into` = (out, y) of an earlier call:
    the other part's rows are already there); negative = the measured choice per width (profiles/r06_call22_microbench_swin_mlp.json)."""
    lib = load_library()
    _need_gpu(x, delta, ln_g, ln_b, w1f, b1, w2f, b2)
    C = x.shape[-1]
    M = x.numel() // C
    assert C in SWIN_MLP_WIDTHS and x.dtype == torch.float32 and x.is_contiguous()
    assert delta is None or (delta.dtype == w1f.dtype and delta.is_contiguous() and delta.shape == x.shape)
    assert w1f.numel() == (4 * C // 32 + 2) * (C // 16) * 512 and w2f.numel() == (4 * C // 32) * (C // 16) * 512
    assert w1f.is_contiguous() and w2f.is_contiguous() and w1f.dtype == w2f.dtype == b1.dtype == b2.dtype == ln_g.dtype and w1f.dtype in _H16
    out = torch.empty_like(x) if into is None else into[0]
    y, ng, nb, ne = None, None, None, 0.0
    if next_ln is not None:
        ng, nb, ne = next_ln
        y = torch.empty(x.shape, dtype=w1f.dtype, device=x.device) if into is None else into[1]
    flags = KERNELS["SWIN_MLP2_FLAGS"] if flags is None else int(flags)
    auto = flags < 0
    if auto:
        # table GELU at every width, one pass (no pass / tail split) below C = 384 -- re-measured on the kernels with counted waits (GPU call 22 of
        # round 6, profiles/r06_call22_microbench_swin_mlp.json: C = 96 0.253 ms unsplit / 0.269 split (erf 0.306), C = 192 0.186 / 0.193 (erf, the
        # default since round 3: 0.214), C = 384 0.158 split / 0.207 unsplit)
        flags = 3 if C < 384 else 2
    if f32_operands():
        flags = 0 if auto else flags & ~2     # precise mode: the erf GELU (|error| <= 1.5e-7), not the interpolation table (7e-6); pass / tail split as measured there
    # (round 5's precise mode sent C = 384 through the tail kernel: two stages of both fp32 weight rings were 196 KB.  The split-precise kernel keeps ONE
    # W2 stage at that width -- 150 KB -- and four waves per workgroup: csrc/swin_mlp2.hip W2ONE.)
    fn = _fn(lib, "mq_swin_mlp2_fwd", w1f)

    def call(fl):
        _chk(fn(_ptr(x), _ptr(delta), _ptr(ln_g), _ptr(ln_b), float(eps), _ptr(w1f), _ptr(b1), _ptr(w2f), _ptr(b2),
                _ptr(out), _ptr(ng), _ptr(nb), float(ne), _ptr(y), M, C, fl, _stream()), "mq_swin_mlp2_fwd")
    side = _tail_stream(x.device) if (KERNELS["SWIN_MLP_TAIL_STREAM"] == 1 and x.is_cuda and not (flags & 29) and _TIMING is None) else None
    with _timed(f"swin_mlp_c{C}", M * C * (4 + 4 + (2 if delta is not None else 0) + (2 if y is not None else 0))):
        if side is None:
            call(flags)
        else:
            # the few blocks beyond the last full pass of the chip (tail kernel) BESIDE the main kernel instead of in front of it: a
            # fork from the current stream (inside a HIP-graph capture: a parallel branch); the two parts touch disjoint rows
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                call(flags | 16)
            call(flags | 8)
            main.wait_stream(side)
    return (out, y) if y is not None else out


_TAIL_STREAMS = {}


def _tail_stream(device):
    s = _TAIL_STREAMS.get(device.index)
    if s is None:
        s = _TAIL_STREAMS[device.index] = torch.cuda.Stream(device=device)
    return s


def conv3x3(x_nhwc, w_packed, bias, n_out, stride=1):
    """x [B,H,W,C] fp16 (NHWC, contiguous H,W,C; any batch stride), w_packed [32|256, 9*C] fp16 (k = tap*C + c),
    bias [n_out] fp16 -> [B, Ho, Wo, n_out] fp16."""
    lib = load_library()
    _need_gpu(x_nhwc, w_packed, bias)
    B, H, W, C = x_nhwc.shape
    assert x_nhwc.dtype in _H16 and x_nhwc.stride(3) == 1 and x_nhwc.stride(2) == C and x_nhwc.stride(1) == W * C
    assert w_packed.is_contiguous() and w_packed.shape == (32 if n_out <= 32 else 256, 9 * C) and w_packed.dtype == x_nhwc.dtype
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    ld = n_out if n_out % 8 == 0 else (n_out + 7) // 8 * 8
    out = torch.empty(B, Ho, Wo, ld, dtype=x_nhwc.dtype, device=x_nhwc.device)
    with _timed("conv3x3"):
        _chk(_fn(lib, "mq_conv3x3_fwd", x_nhwc)(_ptr(x_nhwc), _ptr(w_packed), _ptr(bias), _ptr(out), B, H, W, C, x_nhwc.stride(0), n_out, ld,
                                stride, _stream()), "mq_conv3x3_fwd")
    return out if ld == n_out else out[..., :n_out]


def conv3x3_nchw32(x_nhwc, w_packed, bias, n_out):
    """x [B,H,W,C] fp16 (NHWC rows contiguous, any batch stride), w_packed [32, 9*C] fp16, bias [n_out] fp16
    -> [B, n_out, H, W] fp32 (stride 1, pad 1; n_out <= 32): the DyConv offset / mask conv."""
    lib = load_library()
    _need_gpu(x_nhwc, w_packed, bias)
    B, H, W, C = x_nhwc.shape
    assert x_nhwc.dtype in _H16 and x_nhwc.stride(3) == 1 and x_nhwc.stride(2) == C and x_nhwc.stride(1) == W * C
    assert w_packed.is_contiguous() and w_packed.shape == (32, 9 * C) and w_packed.dtype == x_nhwc.dtype and n_out <= 32
    out = torch.empty(B, n_out, H, W, dtype=torch.float32, device=x_nhwc.device)
    with _timed("conv3x3_small", B * H * W * C * 2 + out.numel() * 4):
        # OFFSET_CONV_VARIANT >= 2: unconditional / in-flight loads (csrc/conv_small2.hip), same results bit for bit
        name = "mq_conv3x3_nchw32_v2_fwd" if KERNELS["OFFSET_CONV_VARIANT"] >= 2 else "mq_conv3x3_nchw32_fwd"
        _chk(_fn(lib, name, x_nhwc)(_ptr(x_nhwc), _ptr(w_packed), _ptr(bias), _ptr(out), B, H, W, C, x_nhwc.stride(0), n_out, _stream()), name)
    return out


class _ConvLevel(ctypes.Structure):
    """mq_conv_level of include/mqdet_hip.h."""
    _fields_ = [("x", _vp), ("out", _vp), ("x_bs", _l), ("H", _i), ("W", _i)]


def conv3x3_nchw32_group_supported(levels, n_out):
    """Shapes mq_conv3x3_nchw32_group_fwd takes (else: conv3x3_nchw32 per level)."""
    if f32_operands() == 1:
        return False          # precise mode: the 180-pixel window of all 256 channels is 196 KB at fp32 -- the per-level kernel (two channel passes) fits
    return 0 < len(levels) <= 8 and n_out <= 32 and all(x.shape[3] == 256 and x.shape[1] * x.shape[2] * 256 < 2 ** 31 for x in levels)


def conv3x3_nchw32_group(levels, w_packed, bias, n_out):
    """conv3x3_nchw32 of every pyramid level with the SAME weights in one launch (csrc/conv_small3.hip): levels = list of x [B,H,W,256]
    16-bit NHWC views -> list of [B, n_out, H, W] fp32.  Equal to the per-level operator up to fp32 summation order."""
    lib = load_library()
    _need_gpu(w_packed, bias, *levels)
    assert conv3x3_nchw32_group_supported(levels, n_out)
    B, C = levels[0].shape[0], levels[0].shape[3]
    assert w_packed.is_contiguous() and w_packed.shape == (32, 9 * C) and w_packed.dtype == levels[0].dtype
    arr = (_ConvLevel * len(levels))()
    outs, nbytes = [], 0
    for a, x in zip(arr, levels):
        Bx, H, W, Cx = x.shape
        assert Bx == B and Cx == C and x.dtype == levels[0].dtype and x.stride(3) == 1 and x.stride(2) == C and x.stride(1) == W * C
        out = torch.empty(B, n_out, H, W, dtype=torch.float32, device=x.device)
        a.x, a.out, a.x_bs, a.H, a.W = x.data_ptr(), out.data_ptr(), x.stride(0), H, W
        outs.append(out)
        nbytes += B * H * W * C * 2 + out.numel() * 4
    with _timed("conv3x3_group", nbytes):
        _chk(_fn(lib, "mq_conv3x3_nchw32_group_fwd", levels[0])(ctypes.cast(arr, _vp), len(levels), _ptr(w_packed), _ptr(bias), B, C, n_out, _stream()),
             "mq_conv3x3_nchw32_group_fwd")
    return outs


def dcnv2(x_nhwc, om, w_packed, bias, stride, want_stats=False, wy=None, wx=None, mask_prob=False, tag="dcnv2_fused", plain=False):
    """Fused DCNv2: x [B,H,W,C] fp16 NHWC, om [B,27,oH,oW] fp32, w_packed [256, 9*C] -> y [B, Ho*Wo, 256] fp16, (Ho, Wo)
    (, sums [B, nblk, 256, 3] fp32 = per-patch GroupNorm / scale-attention statistics of y when want_stats; wy [Ho] /
    wx [Wo] fp32 weight the third statistic, None -> 1/(Ho*Wo))."""
    lib = load_library()
    _need_gpu(x_nhwc, om, w_packed, bias, wy, wx)
    B, H, W, C = x_nhwc.shape
    assert x_nhwc.dtype in _H16 and x_nhwc.stride(3) == 1 and x_nhwc.stride(2) == C and x_nhwc.stride(1) == W * C
    assert om.is_contiguous() and om.dtype == torch.float32 and om.shape[1] == 27
    assert w_packed.is_contiguous() and w_packed.shape == (256, 9 * C) and w_packed.dtype == x_nhwc.dtype
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    y = torch.empty(B, Ho * Wo, 256, dtype=x_nhwc.dtype, device=x_nhwc.device)
    sums = None
    if want_stats:
        sums = torch.empty(B, lib.mq_dcnv2_stats_blocks(H, W, stride), 256, 3, dtype=torch.float32, device=y.device)
        if wy is not None:
            assert wy.dtype == wx.dtype == torch.float32 and wy.numel() == Ho and wx.numel() == Wo
    w_packed, wflag = _dcn_w(w_packed)
    with _timed(tag):
        _chk(_fn(lib, "mq_dcnv2_fwd", x_nhwc)(_ptr(x_nhwc), _ptr(om), _ptr(w_packed), _ptr(bias), _ptr(y), _ptr(sums), _ptr(wy), _ptr(wx),
                              B, H, W, C, x_nhwc.stride(0), om.shape[2], om.shape[3], 256, 256, stride, int(bool(mask_prob)) | (2 if plain else 0) | wflag,
                              _stream()), "mq_dcnv2_fwd")
    return (y, (Ho, Wo), sums) if want_stats else (y, (Ho, Wo))


_DCN_TILED = {}          # id(tensor) -> weak reference: weights that dcn_weight_tiles() produced (flags bit 2 of the DCNv2 entry points)


def dcn_weight_tiles(w):
    """DCNv2 / 3x3 conv weights [256, 9 C] (k = tap * C + c) -> the SAME shape and dtype in LDS-TILE ORDER (KERNELS["DCN_BDMA"], flags bit 2 of
    mq_dcnv2_*): per k-step ks = slice * 9 + tap one 32 KB block that is the byte image of the kernel's B tile, copied global -> LDS by LDS-DMA --
    16-bit builds: 256 rows x 64 channels with the 16-byte chunk c of row r at position c ^ (r & 7); split-precise build: [hi plane | lo plane] of
    256 rows x 32 channels (fp16; x = hi + lo / 2^11).  Done once, when the model is packed."""
    import weakref
    N, K = w.shape
    C = K // 9
    assert N == 256 and K == 9 * C and C % 128 == 0 and w.is_contiguous()
    if w.dtype == torch.float32:
        blk = w.view(256, 9, C // 32, 32).permute(2, 1, 0, 3).reshape(-1, 256 * 32)                 # [ks][row][32]
        hi = blk.to(torch.float16)
        lo = ((blk - hi.float()) * 2048.0).to(torch.float16)
        out = torch.cat([hi, lo], 1).contiguous().view(torch.float32).reshape(256, K)
    else:
        blk = w.view(256, 9, C // 64, 8, 8).permute(2, 1, 0, 3, 4).reshape(-1, 256, 8, 8)             # [ks][row][chunk][8]
        r = torch.arange(256, device=w.device)[:, None]
        pos = torch.arange(8, device=w.device)[None, :]
        src = (pos ^ (r & 7))                                                                        # position p of row r holds chunk p ^ (r & 7)
        out = torch.gather(blk, 2, src[None, :, :, None].expand(blk.shape[0], -1, -1, 8)).reshape(256, K).contiguous()
    if len(_DCN_TILED) > 4096:
        for k in [k for k, ref in _DCN_TILED.items() if ref() is None]:
            del _DCN_TILED[k]
    _DCN_TILED[id(out)] = weakref.ref(out)
    return out


def dcn_weight_rows(wt):
    """Inverse of dcn_weight_tiles: LDS-tile order -> row-major [256, 9 C] (k = tap * C + c).  16-bit dtypes: exact; float32 (the split-precise
    planes): hi + lo / 2^11, the value the kernel multiplies with.  Used by the torch-level emulation of the tests and for inspection."""
    N, K = wt.shape
    C = K // 9
    assert N == 256 and K == 9 * C and C % 128 == 0
    if wt.dtype == torch.float32:
        planes = wt.contiguous().view(torch.float16).reshape(-1, 2, 256, 32)                        # [ks][hi | lo][row][32]
        blk = planes[:, 0].float() + planes[:, 1].float() / 2048.0
        return blk.view(C // 32, 9, 256, 32).permute(2, 1, 0, 3).reshape(256, K).contiguous()
    blk = wt.contiguous().view(-1, 256, 8, 8)                                                      # [ks][row][position][8]
    r = torch.arange(256, device=wt.device)[:, None]
    pos = torch.arange(8, device=wt.device)[None, :]
    chunk = torch.gather(blk, 2, (pos ^ (r & 7))[None, :, :, None].expand(blk.shape[0], -1, -1, 8))   # chunk c sits at position c ^ (r & 7)
    return chunk.view(C // 64, 9, 256, 8, 8).permute(2, 1, 0, 3, 4).reshape(256, K).contiguous()


def dcn_is_tiled(w):
    """Is `w` a tensor dcn_weight_tiles returned (recognised by identity)?"""
    ref = _DCN_TILED.get(id(w))
    return ref is not None and ref() is w


def dcn_bdma():
    """Does the active selection stream the DCNv2 weights by LDS-DMA?  KERNELS["DCN_BDMA"]: 1 always (default), 0 never, -1 in the split-precise mode only."""
    k = KERNELS.get("DCN_BDMA", 1)
    return k == 1 or (k == -1 and bool(f32_operands()))


def _dcn_w(w):
    """(weights to hand to the kernel, flags bit 2).  Tile-ordered weights are recognised by identity; with KERNELS["DCN_BDMA"] = 1 row-major
    weights are re-ordered per call (tests, the operator-level wrappers: the pipeline packs once)."""
    ref = _DCN_TILED.get(id(w))
    if ref is not None and ref() is w:
        return w, 4
    if dcn_bdma():
        return dcn_weight_tiles(w), 4
    return w, 0


class _DcnBranch(ctypes.Structure):
    """mq_dcn_branch of include/mqdet_hip.h."""
    _fields_ = [(n, _vp) for n in ("x", "om", "w", "bias", "out", "stats", "wy", "wx")] + [("x_bs", _l)] + \
               [(n, _i) for n in ("B", "H", "W", "C", "oH", "oW", "N", "out_ld", "stride", "flags")]


def dcnv2_group(branches, want_stats=True, tag="dcnv2_fused", ablation=0):
    """ONE launch for several DCNv2 calls (mq_dcnv2_group_fwd).  branches: list of dicts with x [B,H,W,C] fp16 NHWC view,
    om [B,27,oH,oW] fp32, w [256, 9*C] fp16, bias [256] fp16, stride, wy / wx (or None)
    -> list of (y [B, Ho*Wo, 256] fp16, (Ho, Wo), sums or None) in the same order."""
    lib = load_library()
    arr = (_DcnBranch * len(branches))()
    outs = []
    keep = []                                                 # per-call re-ordered weights stay alive until the launch is issued
    for i, (a, br) in enumerate(zip(arr, branches)):
        x, om, w, bias, stride = br["x"], br["om"], br["w"], br["bias"], br["stride"]
        wy, wx = br.get("wy"), br.get("wx")
        _need_gpu(x, om, w, bias, wy, wx)
        B, H, W, C = x.shape
        assert x.dtype in _H16 and x.stride(3) == 1 and x.stride(2) == C and x.stride(1) == W * C
        assert om.is_contiguous() and om.dtype == torch.float32 and om.shape[1] == 27
        assert w.is_contiguous() and w.shape == (256, 9 * C) and w.dtype == x.dtype
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        y = torch.empty(B, Ho * Wo, 256, dtype=x.dtype, device=x.device)
        sums = None
        if want_stats:
            sums = torch.empty(B, lib.mq_dcnv2_stats_blocks(H, W, stride), 256, 3, dtype=torch.float32, device=y.device)
            if wy is not None:
                assert wy.dtype == wx.dtype == torch.float32 and wy.numel() == Ho and wx.numel() == Wo
        w, wflag = _dcn_w(w)
        keep.append(w)
        a.x, a.om, a.w, a.bias, a.out = x.data_ptr(), om.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr()
        a.stats = sums.data_ptr() if sums is not None else None
        a.wy = wy.data_ptr() if wy is not None else None
        a.wx = wx.data_ptr() if wx is not None else None
        a.x_bs, a.B, a.H, a.W, a.C, a.oH, a.oW = x.stride(0), B, H, W, C, om.shape[2], om.shape[3]
        a.N, a.out_ld, a.stride, a.flags = 256, 256, stride, (int(bool(br.get("mask_prob", False))) | (2 if br.get("plain", False) else 0) | wflag
                                                             | ((int(ablation) & 15) << 8 if i == 0 else 0))
        outs.append((y, (Ho, Wo), sums))
    with _timed(tag):
        _chk(_fn(lib, "mq_dcnv2_group_fwd", *[br["x"] for br in branches])(ctypes.cast(arr, _vp), len(branches), _stream()), "mq_dcnv2_group_fwd")
    return outs


def dyconv_branch_coef(y, Wsrc, gamma, beta, attn_w, attn_b, groups, eps, nbranches, wy=None, wx=None, sums=None):
    """y [B,n,C] fp16 (DCN output of one branch) -> coef [B,C,2] fp32 (GN affine x scale attention / nbranches).
    sums: statistics already produced by the fused DCNv2 kernel ([B, nblk, C, 3]); None -> one pass over y here."""
    lib = load_library()
    _need_gpu(y, gamma, beta, attn_w, attn_b, wy, wx)
    B, n, C = y.shape
    assert y.is_contiguous() and y.dtype in _H16 and gamma.dtype == y.dtype
    assert attn_w.dtype == torch.float32 and attn_b.dtype == torch.float32
    coef = torch.empty(B, C, 2, dtype=torch.float32, device=y.device)
    nblk = 0
    if sums is None:
        sums = torch.empty(B, (n + 255) // 256, C, 3, dtype=torch.float32, device=y.device)
        with _timed("dyconv_stats"):
            _chk(_fn(lib, "mq_dyconv_stats", y)(_ptr(y), _ptr(sums), _ptr(wy), _ptr(wx), B, n, Wsrc, C, _stream()), "mq_dyconv_stats")
    else:
        assert sums.dtype == torch.float32 and sums.is_contiguous() and sums.shape[0] == B and sums.shape[2:] == (C, 3)
        nblk = sums.shape[1]
    _chk(_fn(lib, "mq_dyconv_coef", gamma)(_ptr(sums), _ptr(gamma), _ptr(beta), _ptr(attn_w), _ptr(attn_b), _ptr(coef), B, n, nblk, C, groups,
                            float(eps), nbranches, _stream()), "mq_dyconv_coef")
    return coef


class _CoefBranch(ctypes.Structure):
    """mq_coef_branch of include/mqdet_hip.h."""
    _fields_ = [(n, _vp) for n in ("sums", "gamma", "beta", "coef")] + [(n, _i) for n in ("nblk", "n", "nbranches", "reserved")]


def dyconv_coef_group(items, attn_w, attn_b, groups, eps):
    """Coefficients of several DCN branches in one launch.  items: list of dicts {sums [B,nblk,C,3] fp32, n (positions),
    gamma, beta (fp16 [C]), nbranches} -> list of coef [B,C,2] fp32."""
    lib = load_library()
    arr = (_CoefBranch * len(items))()
    outs = []
    B = C = None
    for a, it in zip(arr, items):
        sums = it["sums"]
        _need_gpu(sums, it["gamma"], it["beta"], attn_w, attn_b)
        assert sums.dtype == torch.float32 and sums.is_contiguous() and sums.shape[2:] == (256, 3)
        B, C = sums.shape[0], sums.shape[2]
        coef = torch.empty(B, C, 2, dtype=torch.float32, device=sums.device)
        a.sums, a.gamma, a.beta, a.coef = sums.data_ptr(), it["gamma"].data_ptr(), it["beta"].data_ptr(), coef.data_ptr()
        a.nblk, a.n, a.nbranches, a.reserved = sums.shape[1], int(it["n"]), int(it["nbranches"]), 0
        outs.append(coef)
    _chk(_fn(lib, "mq_dyconv_coef_group", *[it["gamma"] for it in items])(ctypes.cast(arr, _vp), len(items), _ptr(attn_w), _ptr(attn_b), B, C, groups, float(eps), _stream()),
         "mq_dyconv_coef_group")
    return outs


def dyconv_fuse(branches, H, W, out=None):
    """branches: list of (y [B,hs*ws,C], coef [B,C,2], hs, ws) -> out [B,H*W,C] fp16, pool [B,C] fp32.
    out: optional destination view [B,H*W,C] with contiguous rows and any batch stride (a level's slice of the pyramid
    token buffer)."""
    lib = load_library()
    y0 = branches[0][0]
    B, _, C = y0.shape
    if out is None:
        out = torch.empty(B, H * W, C, dtype=y0.dtype, device=y0.device)
    assert out.shape == (B, H * W, C) and out.dtype == y0.dtype and y0.dtype in _H16 and out.stride(2) == 1 and out.stride(1) == C
    pool = torch.empty(B, (H * W + 127) // 128, C, dtype=torch.float32, device=y0.device)
    args = []
    for k in range(3):
        if k < len(branches):
            y, cf, hs, ws = branches[k]
            _need_gpu(y, cf)
            assert y.is_contiguous() and cf.is_contiguous() and y.shape[1] == hs * ws
            args += [_ptr(y), _ptr(cf), hs, ws]
        else:
            args += [_ptr(None), _ptr(None), 0, 0]
    with _timed("dyconv_fuse", sum(b_[0].numel() * 2 for b_ in branches) + out.shape[0] * out.shape[1] * C * 2):
        _chk(_fn(lib, "mq_dyconv_fuse", *[b_[0] for b_ in branches])(*args, len(branches), _ptr(out), out.stride(0), _ptr(pool), B, H, W, C, _stream()), "mq_dyconv_fuse")
    return out, pool


class _FuseLevel(ctypes.Structure):
    """mq_fuse_level of include/mqdet_hip.h."""
    _fields_ = [("y", _vp * 3), ("coef", _vp * 3), ("hs", _i * 3), ("ws", _i * 3), ("nbranches", _i), ("H", _i), ("W", _i), ("reserved", _i),
                ("out", _vp), ("out_bs", _l), ("pool", _vp), ("relu_coef", _vp)]


def dyconv_epilogue_group(levels, w0, b0, w2, b2, relu_coef):
    """dyconv_fuse + dyrelu_coef of every pyramid level of a DyConv layer in two launches (mq_dyconv_epilogue_group).
    levels: list of (branches, H, W, out) with the arguments of dyconv_fuse; relu_coef [NL, B, 4, C] fp32 receives the DYReLU
    coefficients of every level.  Same arithmetic as the per-level calls (equal results)."""
    lib = load_library()
    _need_gpu(w0, b0, w2, b2, relu_coef)
    y0 = levels[0][0][0][0]
    B, _, C = y0.shape
    assert 0 < len(levels) <= 8 and C == 256 and relu_coef.shape == (len(levels), B, 4, C) and relu_coef.is_contiguous() and relu_coef.dtype == torch.float32
    assert w0.is_contiguous() and w2.is_contiguous() and w0.dtype == y0.dtype and y0.dtype in _H16
    arr = (_FuseLevel * len(levels))()
    keep, nbytes = [], 0
    for l, (a, (branches, H, W, out)) in enumerate(zip(arr, levels)):
        assert out.shape == (B, H * W, C) and out.dtype == y0.dtype and out.stride(2) == 1 and out.stride(1) == C and 1 <= len(branches) <= 3
        pool = torch.empty(B, (H * W + 127) // 128, C, dtype=torch.float32, device=y0.device)
        keep.append(pool)
        for k, (y, cf, hs, ws) in enumerate(branches):
            _need_gpu(y, cf)
            assert y.is_contiguous() and cf.is_contiguous() and y.shape == (B, hs * ws, C) and y.dtype == y0.dtype and cf.dtype == torch.float32
            a.y[k], a.coef[k], a.hs[k], a.ws[k] = y.data_ptr(), cf.data_ptr(), hs, ws
            nbytes += y.numel() * 2
        a.nbranches, a.H, a.W, a.reserved = len(branches), H, W, 0
        a.out, a.out_bs, a.pool, a.relu_coef = out.data_ptr(), out.stride(0), pool.data_ptr(), relu_coef[l].data_ptr()
        nbytes += B * H * W * C * 2
    with _timed("dyconv_epilogue_group", nbytes):
        _chk(_fn(lib, "mq_dyconv_epilogue_group", y0)(ctypes.cast(arr, _vp), len(levels), _ptr(w0), _ptr(b0), _ptr(w2), _ptr(b2), B, C, _stream()),
             "mq_dyconv_epilogue_group")
    return relu_coef


def dyrelu_(x, pool, w0, b0, w2, b2):
    """In-place DYReLU on x [B,n,C] fp16 (contiguous rows, any batch stride) given pool [B,C] = sum over positions of x."""
    lib = load_library()
    _need_gpu(x, pool, w0, b0, w2, b2)
    B, n, C = x.shape
    assert x.stride(2) == 1 and x.stride(1) == C and w0.is_contiguous() and w2.is_contiguous() and w0.dtype == x.dtype and x.dtype in _H16
    coef = torch.empty(B, 4, C, dtype=torch.float32, device=x.device)
    _chk(_fn(lib, "mq_dyrelu_coef", w0)(_ptr(pool), _ptr(w0), _ptr(b0), _ptr(w2), _ptr(b2), _ptr(coef), B, n, C, _stream()),
         "mq_dyrelu_coef")
    with _timed("dyrelu_apply", 2 * B * n * C * 2):
        _chk(_fn(lib, "mq_dyrelu_apply", x)(_ptr(x), _ptr(coef), B, n, C, x.stride(0), _stream()), "mq_dyrelu_apply")
    return x


def dyrelu_coef(pool, n, w0, b0, w2, b2, out=None):
    """DYReLU coefficients (mq_dyrelu_coef): pool [B, ceil(n / 128), C] fp32 = per-block sums over the n positions of a level
    (dyconv_fuse's second result) -> [B,4,C] fp32 (a1, b1, a2, b2)."""
    lib = load_library()
    _need_gpu(pool, w0, b0, w2, b2)
    B, C = pool.shape[0], pool.shape[-1]
    assert pool.is_contiguous() and pool.dtype == torch.float32 and pool.numel() == B * ((int(n) + 127) // 128) * C
    coef = torch.empty(B, 4, C, dtype=torch.float32, device=pool.device) if out is None else out
    assert coef.shape == (B, 4, C) and coef.is_contiguous() and coef.dtype == torch.float32 and w0.is_contiguous() and w2.is_contiguous()
    _chk(_fn(lib, "mq_dyrelu_coef", w0)(_ptr(pool), _ptr(w0), _ptr(b0), _ptr(w2), _ptr(b2), _ptr(coef), B, int(n), C, _stream()), "mq_dyrelu_coef")
    return coef


def add_upsample_nearest_(dst, src):
    """FPN top-down step in place (mq_add_upsample_nearest): dst [B,H,W,C] += nearest-up-sampled src [B,Hc,Wc,C], both NHWC 16-bit
    contiguous = dst + F.interpolate(src, size=(H, W), mode="nearest") with one rounding."""
    lib = load_library()
    _need_gpu(dst, src)
    B, H, W, C = dst.shape
    assert src.shape[0] == B and src.shape[3] == C and dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.dtype in _H16
    with _timed("fpn_topdown", 2 * dst.numel() * 2 + src.numel() * 2):
        _chk(_fn(lib, "mq_add_upsample_nearest", dst)(_ptr(dst), _ptr(src), B, H, W, src.shape[1], src.shape[2], C, _stream()),
             "mq_add_upsample_nearest")
    return dst


def pool2x2_tokens_supported(feats):
    """Shapes / layouts mq_pool2x2_tokens_fwd takes: up to 8 levels, [B,C,H,W] VIEWS of NHWC storage (what the FPN returns), every level at least
    2 x 2, C a multiple of 8 -- anything else goes through the torch statement."""
    if not (0 < len(feats) <= 8):
        return False
    B, C = feats[0].shape[0], feats[0].shape[1]
    for f in feats:
        if f.dim() != 4 or f.shape[0] != B or f.shape[1] != C or f.dtype != feats[0].dtype or f.shape[2] < 2 or f.shape[3] < 2:
            return False
        if f.stride(1) != 1 or f.stride(3) != C or f.stride(2) != f.shape[3] * C or f.data_ptr() % 16 or f.stride(0) % 8:
            return False
    return C % 8 == 0 and feats[0].dtype in (torch.float16, torch.bfloat16, torch.float32)


def pool2x2_tokens(feats):
    """Pooled FPN tokens of the GCP pre-select in one launch (mq_pool2x2_tokens_fwd): feats = list of [B,C,H,W] views of NHWC tensors (what the FPN
    returns) -> [B, sum (H/2)*(W/2), C] = torch.cat([F.avg_pool2d(f, 2).permute(0, 2, 3, 1).flatten(1, 2) for f in feats], 1), bit for bit."""
    lib = load_library()
    _need_gpu(*feats)
    assert 0 < len(feats) <= 8
    B, C = feats[0].shape[0], feats[0].shape[1]
    arr = (_ConvLevel * len(feats))()
    n = 0
    for a, f in zip(arr, feats):
        x = f.permute(0, 2, 3, 1)
        Bx, H, W, Cx = x.shape
        assert Bx == B and Cx == C and x.dtype == feats[0].dtype and x.stride(3) == 1 and x.stride(2) == C and x.stride(1) == W * C and H >= 2 and W >= 2
        a.x, a.out, a.x_bs, a.H, a.W = x.data_ptr(), None, x.stride(0), H, W
        n += (H // 2) * (W // 2)
    out = torch.empty(B, n, C, dtype=feats[0].dtype, device=feats[0].device)
    with _timed("pool2x2_tokens", sum(f.numel() for f in feats) * 2 + out.numel() * 2):
        _chk(_fn(lib, "mq_pool2x2_tokens_fwd", feats[0])(ctypes.cast(arr, _vp), len(feats), _ptr(out), B, C, _stream()), "mq_pool2x2_tokens_fwd")
    return out


def dyrelu_apply_(x, coef):
    """In-place DYReLU on x [B,n,C] 16-bit (contiguous rows, any batch stride) with given coefficients [B,4,C] (mq_dyrelu_apply)."""
    lib = load_library()
    _need_gpu(x, coef)
    B, n, C = x.shape
    assert x.stride(2) == 1 and x.stride(1) == C and x.dtype in _H16 and coef.shape == (B, 4, C) and coef.is_contiguous()
    _chk(_fn(lib, "mq_dyrelu_apply", x)(_ptr(x), _ptr(coef), B, n, C, x.stride(0), _stream()), "mq_dyrelu_apply")
    return x


def dyrelu_layer_norm(x, coef, sizes, gamma, beta, eps):
    """LayerNorm(DYReLU(x)) (mq_dyrelu_ln_fwd): x [B,N,256] 16-bit (rows contiguous), coef [NL,B,4,256] fp32 (dyrelu_coef per level),
    sizes = [(H, W)] of the NL levels whose tokens are concatenated along N -> [B,N,256] 16-bit."""
    lib = load_library()
    _need_gpu(x, coef, gamma, beta)
    B, N, C = x.shape
    NL = len(sizes)
    offs = [0]
    for (h, w) in sizes:
        offs.append(offs[-1] + int(h) * int(w))
    assert C == 256 and offs[-1] == N and x.stride(2) == 1 and x.stride(1) == C and x.dtype in _H16 and gamma.dtype == x.dtype
    assert coef.shape == (NL, B, 4, C) and coef.is_contiguous() and coef.dtype == torch.float32
    y = torch.empty(B, N, C, dtype=x.dtype, device=x.device)
    rf = (ctypes.c_int * (NL + 1))(*offs)
    with _timed("layernorm_c256_dyrelu", 2 * x.numel() * 2):
        _chk(_fn(lib, "mq_dyrelu_ln_fwd", x)(_ptr(x), x.stride(0), _ptr(coef), ctypes.cast(rf, _vp), NL, _ptr(gamma), _ptr(beta), float(eps),
                                             _ptr(y), B, N, C, _stream()), "mq_dyrelu_ln_fwd")
    return y


SCORE_AGG = {"MEAN": 0, "MAX": 1, "POWER": 2, "ONEHOT": 0}      # ONEHOT: MEAN over the one-token index of token_index_onehot


def align_scores(dot, tbias, tokidx, ctr, thr, want_cls=False, agg=0):
    """dot [B,HW,T] fp16 or fp32 (contiguous rows, any batch stride), tbias [B,T] fp32, tokidx [L,MT] (one caption for the
    batch) or [B,L,MT] (one per item) int32, ctr [B,HW] fp16 -> ranked [B,HW,L] fp32 (, cls)."""
    lib = load_library()
    _need_gpu(dot, tbias, tokidx, ctr)
    B, HW, T = dot.shape
    L, MT = tokidx.shape[-2:]
    assert dot.stride(2) == 1 and dot.stride(1) == T and dot.dtype in (ctr.dtype, torch.float32)
    assert ctr.dtype in _H16 and ctr.is_contiguous()
    assert tbias.dtype == torch.float32 and tbias.is_contiguous() and tokidx.dtype == torch.int32 and tokidx.is_contiguous()
    assert tokidx.dim() == 2 or tokidx.shape[0] == B
    out = torch.empty(B, HW, L, dtype=torch.float32, device=dot.device)
    cls = torch.empty_like(out) if want_cls else None
    with _timed("align_scores", B * HW * T * dot.element_size() + out.numel() * 4):
        _chk(_fn(lib, "mq_align_scores_fwd", ctr)(_ptr(dot), int(dot.dtype == torch.float32), _ptr(tbias), _ptr(tokidx),
                                     L * MT if tokidx.dim() == 3 else 0, _ptr(ctr), _ptr(out), _ptr(cls), B, HW, T, L, MT,
                                     float(thr), dot.stride(0), int(agg), _stream()), "mq_align_scores_fwd")
    return (out, cls) if want_cls else out


def align_fused(tok, tk, tbias, wbc, bbc, scales, tokidx, sizes, thr, agg=0, kv_max=0, want_cls=False, want_logits=False):
    """Prediction heads + region-word alignment + per-location scoring for ALL pyramid levels in one launch (mq_align_fused_fwd).
    tok [B,N,256] 16-bit (levels concatenated in `sizes` order), tk [B,T,256] 16-bit (projected text tokens / exp(log_scale)), tbias [B,T]
    fp32, wbc [16,256] 16-bit / bbc [8] fp32 (box + centerness rows), scales [NL] fp32, tokidx [L,MT] or [B,L,MT] int32, sizes [(H, W)],
    kv_max: host-side upper bound of the live text tokens (0 = T).
    -> dict: ranked [NL x [B,HW,L] fp32], reg [NL x [B,HW,4] fp32], ctr [B,N] fp32 (centerness logits), cls (want_cls), logits
    [B,N,T] fp32 dot products without the bias (want_logits; columns beyond the live text blocks are zero)."""
    lib = load_library()
    _need_gpu(tok, tk, tbias, wbc, bbc, scales, tokidx)
    B, N, C = tok.shape
    T = tk.shape[1]
    L, MT = tokidx.shape[-2:]
    NL = len(sizes)
    offs = [0]
    for (h, w) in sizes:
        offs.append(offs[-1] + int(h) * int(w))
    assert C == 256 and offs[-1] == N and tk.shape == (B, T, 256) and tok.is_contiguous() and tk.is_contiguous()
    assert tok.dtype == tk.dtype == wbc.dtype and tok.dtype in _H16 and wbc.shape == (16, 256) and wbc.is_contiguous()
    assert tbias.dtype == torch.float32 and tbias.shape == (B, T) and tbias.is_contiguous()
    assert bbc.dtype == scales.dtype == torch.float32 and bbc.numel() >= 8 and scales.numel() >= NL
    assert tokidx.dtype == torch.int32 and tokidx.is_contiguous() and (tokidx.dim() == 2 or tokidx.shape[0] == B)
    live = kv_max if 0 < kv_max < T else T
    if f32_operands() == 1 and -(-live // 16) * 16 > 144:
        raise RuntimeError(f"mq_align_fused_fwd_f32: the text tile of {live} live tokens does not fit the LDS at fp32 (up to 144; the GEMM path takes longer captions)")
    dev = tok.device
    ranked = torch.empty(B * N * L, dtype=torch.float32, device=dev)
    cls = torch.empty(B * N * L, dtype=torch.float32, device=dev) if want_cls else None
    reg = torch.empty(B * N * 4, dtype=torch.float32, device=dev)
    ctr = torch.empty(B, N, dtype=torch.float32, device=dev)
    logits = torch.zeros(B, N, T, dtype=torch.float32, device=dev) if want_logits else None
    lvl = (ctypes.c_int * (NL + 1))(*offs)
    with _timed("align_fused", tok.numel() * 2 + ranked.numel() * 4 + reg.numel() * 4):
        _chk(_fn(lib, "mq_align_fused_fwd", tok)(_ptr(tok), _ptr(tk), _ptr(tbias), _ptr(wbc), _ptr(bbc), _ptr(scales), _ptr(tokidx),
                                                 L * MT if tokidx.dim() == 3 else 0, ctypes.cast(lvl, _vp), _ptr(ranked), _ptr(cls), _ptr(reg),
                                                 _ptr(ctr), _ptr(logits), B, N, T, int(kv_max), L, MT, NL, float(thr), int(agg), _stream()),
             "mq_align_fused_fwd")

    def levels(flat, width):
        return [flat[offs[l] * B * width:offs[l + 1] * B * width].view(B, offs[l + 1] - offs[l], width) for l in range(NL)]
    out = {"ranked": levels(ranked, L), "reg": levels(reg, 4), "ctr": ctr}
    if want_cls:
        out["cls"] = levels(cls, L)
    if want_logits:
        out["logits"] = logits
    return out


def post_select_supported(hws, ks, B, L):
    """Can mq_post_select_fwd take these level sizes?  (host-side query of its slicing plan; no device work)"""
    NL = len(hws)
    if NL < 1 or NL > 8:
        return False
    hw = (ctypes.c_int * NL)(*[int(h) for h in hws])
    kk = (ctypes.c_int * NL)(*[int(k) for k in ks])
    return load_library().mq_post_select_workspace_bytes(ctypes.cast(hw, _vp), ctypes.cast(kk, _vp), NL, int(B), int(L)) >= 0


def post_select(ranked, reg, anchors, ks, label_ids, im_wh):
    """Per (image, level) the ks[l] best candidates (value > 0) of ranked[l] [B,HW_l,L] fp32, decoded with reg[l] [B,HW_l,4] fp32 and
    anchors[l] [HW_l,4] (mq_post_select_fwd).  -> boxes [B,tot,4], scores [B,tot] (-1 = empty slot), labels, ids [B,tot] int32; level l owns
    the slots [sum ks[:l], sum ks[:l+1]), sorted by (score desc, flat index asc), empty slots last."""
    lib = load_library()
    NL = len(ranked)
    B, _, L = ranked[0].shape
    dev = ranked[0].device
    _need_gpu(*ranked, *reg, *anchors, label_ids, im_wh)
    for r, g, a in zip(ranked, reg, anchors):
        assert r.dtype == g.dtype == a.dtype == torch.float32 and r.is_contiguous() and g.is_contiguous() and a.is_contiguous()
        assert r.shape[0] == B and r.shape[2] == L and g.shape == (B, r.shape[1], 4) and a.shape == (r.shape[1], 4)
    assert label_ids.dtype == torch.int32 and label_ids.is_contiguous() and im_wh.dtype == torch.float32 and im_wh.is_contiguous()
    tot = int(sum(ks))
    boxes = torch.empty(B, tot, 4, dtype=torch.float32, device=dev)
    scores = torch.empty(B, tot, dtype=torch.float32, device=dev)
    labels = torch.empty(B, tot, dtype=torch.int32, device=dev)
    ids = torch.empty(B, tot, dtype=torch.int32, device=dev)
    pp = lambda ts: (ctypes.c_void_p * NL)(*[t.data_ptr() for t in ts])      # noqa: E731  host arrays of device pointers
    hw = (ctypes.c_int * NL)(*[int(r.shape[1]) for r in ranked])
    kk = (ctypes.c_int * NL)(*[int(k) for k in ks])
    wsb = lib.mq_post_select_workspace_bytes(ctypes.cast(hw, _vp), ctypes.cast(kk, _vp), NL, B, L)
    if wsb < 0:
        raise RuntimeError("mq_post_select_fwd: level sizes out of range (ops.post_select_supported)")
    ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=dev)
    with _timed("post_select", sum(r.numel() for r in ranked) * 4):
        _chk(lib.mq_post_select_fwd(ctypes.cast(pp(ranked), _vp), ctypes.cast(pp(reg), _vp), ctypes.cast(pp(anchors), _vp), ctypes.cast(hw, _vp),
                                    ctypes.cast(kk, _vp), NL, B, L, _ptr(label_ids), L if label_ids.dim() == 2 else 0, _ptr(im_wh), _ptr(ws), _ptr(boxes),
                                    _ptr(scores), _ptr(labels), _ptr(ids), _stream()), "mq_post_select_fwd")
    return boxes, scores, labels, ids


def post_sort(boxes, scores, labels, ks):
    """Per-level candidate lists [B,tot,...] (list l = ks[l] slots, each sorted by score, as post_select writes them) -> ONE list per image
    ordered by (score desc, level asc, position asc), empty slots last, + nvalid [B] int32 (mq_post_sort_fwd)."""
    lib = load_library()
    _need_gpu(boxes, scores, labels)
    B, tot = scores.shape
    NL = len(ks)
    offs = [0]
    for k in ks:
        offs.append(offs[-1] + int(k))
    assert offs[-1] == tot
    off = (ctypes.c_int * (NL + 1))(*offs)
    bo, so, lo = torch.empty_like(boxes), torch.empty_like(scores), torch.empty_like(labels)
    nvalid = torch.empty(B, dtype=torch.int32, device=scores.device)
    with _timed("post_sort"):
        _chk(lib.mq_post_sort_fwd(_ptr(boxes), _ptr(scores), _ptr(labels), ctypes.cast(off, _vp), NL, _ptr(bo), _ptr(so), _ptr(lo), _ptr(nvalid), B, tot,
                                  _stream()), "mq_post_sort_fwd")
    return bo, so, lo, nvalid


def post_finalize(boxes, scores, labels, keep, K, K2):
    """Score-sorted rows + NMS keep flags [B,tot] uint8 -> packed [B,K2,6] (x1, y1, x2, y2, score, label; unused rows score -1) and
    counts [B] int32 (live rows | 1 << 16 on tie overflow) (mq_post_finalize_fwd)."""
    lib = load_library()
    _need_gpu(boxes, scores, labels, keep)
    B, tot = scores.shape
    assert keep.dtype == torch.uint8 and keep.is_contiguous() and keep.shape == (B, tot)
    out = torch.empty(B, K2, 6, dtype=torch.float32, device=scores.device)
    counts = torch.empty(B, dtype=torch.int32, device=scores.device)
    with _timed("post_finalize"):
        _chk(lib.mq_post_finalize_fwd(_ptr(boxes), _ptr(scores), _ptr(labels), _ptr(keep), _ptr(out), _ptr(counts), B, tot, int(K), int(K2), _stream()),
             "mq_post_finalize_fwd")
    return out, counts


def box_decode(val, flat, reg, anchors, label_ids, im_wh, boxes, scores, labels, HW, L, out_off):
    """Decode top-K candidates of one level into columns [out_off, out_off+K) of the per-image arrays.
    label_ids [L] (shared) or [B, L] int32."""
    lib = load_library()
    _need_gpu(val, flat, reg, anchors, label_ids, im_wh, boxes, scores, labels)
    B, K = val.shape
    assert val.dtype == torch.float32 and flat.dtype == torch.int64 and reg.dtype in _H16 + (torch.float32,)
    assert val.is_contiguous() and flat.is_contiguous() and reg.is_contiguous() and anchors.is_contiguous()
    assert boxes.dtype == torch.float32 and scores.dtype == torch.float32 and labels.dtype == torch.int32
    assert label_ids.is_contiguous() and label_ids.dtype == torch.int32 and (label_ids.dim() == 1 or label_ids.shape[0] == B)
    _chk(_fn(lib, "mq_box_decode", reg)(_ptr(val), _ptr(flat), _ptr(reg), int(reg.dtype == torch.float32), _ptr(anchors), _ptr(label_ids), L if label_ids.dim() == 2 else 0,
                           _ptr(im_wh), _ptr(boxes), _ptr(scores), _ptr(labels), B, K, HW, L, boxes.shape[1], out_off, _stream()),
         "mq_box_decode")


def roi_align(feat, rois, output_size, spatial_scale, sampling_ratio, aligned=True, reduce_mean=False):
    """feat [N,C,H,W] (any strides -- the product's pyramid levels are NHWC memory viewed as NCHW; fp16 or fp32),
    rois [R,5] fp32 (batch index, x1, y1, x2, y2) -> [R,C,PH,PW] fp32, or [R,C] (mean over the bins) when reduce_mean."""
    lib = load_library()
    _need_gpu(feat, rois)
    N, C, H, W = feat.shape
    PH, PW = (output_size, output_size) if isinstance(output_size, int) else output_size
    assert feat.dtype in _H16 + (torch.float32,) and rois.dtype == torch.float32 and rois.shape[1] == 5
    rois = rois.contiguous()
    R = rois.shape[0]
    out = torch.empty((R, C) if reduce_mean else (R, C, PH, PW), dtype=torch.float32, device=feat.device)
    _chk(_fn(lib, "mq_roi_align_fwd", feat)(_ptr(feat), int(feat.dtype == torch.float32), _ptr(rois), _ptr(out), R, C, H, W, feat.stride(0),
                              feat.stride(1), feat.stride(2), feat.stride(3), PH, PW, float(spatial_scale), int(sampling_ratio),
                              int(bool(aligned)), int(bool(reduce_mean)), _stream()), "mq_roi_align_fwd")
    return out


_MSDA_SHAPES = {}


def ms_deform_attn(value, spatial_shapes, sampling_locations, attention_weights, out_dtype=None):
    """value [B,S,M,D] fp16 / fp32 (contiguous), spatial_shapes: list of (H, W) (sum H*W == S), sampling_locations
    [B,Q,M,L,P,2] fp32, attention_weights [B,Q,M,L,P] fp32 -> [B,Q,M*D] in out_dtype (default: value's dtype)."""
    lib = load_library()
    _need_gpu(value, sampling_locations, attention_weights)
    B, S, M, D = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    shapes = tuple((int(h), int(w)) for h, w in spatial_shapes)
    assert len(shapes) == L and sum(h * w for h, w in shapes) == S
    assert value.is_contiguous() and value.dtype in _H16 + (torch.float32,)
    assert sampling_locations.dtype == attention_weights.dtype == torch.float32
    assert sampling_locations.is_contiguous() and attention_weights.is_contiguous() and attention_weights.shape == (B, Q, M, L, P)
    hw, start = _msda_shapes(shapes, value.device)
    out_dtype = out_dtype or value.dtype
    out = torch.empty(B, Q, M * D, dtype=out_dtype, device=value.device)
    nb = value.numel() * value.element_size() + sampling_locations.numel() * 4 + attention_weights.numel() * 4 + out.numel() * out.element_size()
    with _timed(f"msdeform_attn_q{Q}", nb):
        _chk(_fn(lib, "mq_msdeform_attn_fwd", value, out)(_ptr(value), int(value.dtype == torch.float32), _ptr(hw), _ptr(start), _ptr(sampling_locations),
                                      _ptr(attention_weights), _ptr(out), int(out_dtype == torch.float32), B, S, M, D, L, Q, P, _stream()),
             "mq_msdeform_attn_fwd")
    return out


def _msda_shapes(shapes, device):
    key = (shapes, device)
    if key not in _MSDA_SHAPES:                   # [L, 2] (H, W) and level start indices as int64 device tensors, cached
        hw = torch.tensor(shapes, dtype=torch.int64)
        start = torch.cat([hw.new_zeros(1), (hw[:, 0] * hw[:, 1]).cumsum(0)[:-1]])
        _MSDA_SHAPES[key] = (hw.to(device), start.to(device))
    return _MSDA_SHAPES[key]


def ms_deform_attn_q(value, spatial_shapes, qproj, ref, heads, out_dtype=None, valid_hw=None):
    """Fused-query form (mq_msdeform_attn_q_fwd).  value [B,S,>=heads*D] fp16 / fp32 view with unit channel stride (token /
    batch strides free: a column slice of a wider projection is fine), qproj [B,Q,heads*16*3] fp16 contiguous =
    [offsets | logits] of the fused projection, ref [B,Q,4,2|4] fp32 contiguous, valid_hw [B,4,2] int32 or None (un-padded
    rows / columns per level: value rows outside count as zero) -> [B,Q,heads*D]."""
    lib = load_library()
    _need_gpu(value, qproj, ref)
    B, S, C = value.shape
    D = C // heads
    Q = qproj.shape[1]
    shapes = tuple((int(h), int(w)) for h, w in spatial_shapes)
    L, P = len(shapes), 4
    assert sum(h * w for h, w in shapes) == S and value.stride(2) == 1 and value.dtype in _H16 + (torch.float32,)
    assert qproj.dtype in _H16 and qproj.is_contiguous() and qproj.shape == (B, Q, heads * L * P * 3)
    assert ref.dtype == torch.float32 and ref.is_contiguous() and ref.shape[:3] == (B, Q, L) and ref.shape[3] in (2, 4)
    if valid_hw is not None:
        _need_gpu(valid_hw)
        assert valid_hw.dtype == torch.int32 and valid_hw.is_contiguous() and valid_hw.shape == (B, L, 2)
    hw, start = _msda_shapes(shapes, value.device)
    out_dtype = out_dtype or value.dtype
    out = torch.empty(B, Q, C, dtype=out_dtype, device=value.device)
    nb = B * S * C * value.element_size() + qproj.numel() * 2 + ref.numel() * 4 + out.numel() * out.element_size()
    with _timed(f"msdeform_attn_q{Q}", nb):
        _chk(_fn(lib, "mq_msdeform_attn_q_fwd", value, qproj, out)(_ptr(value), int(value.dtype == torch.float32), value.stride(0), value.stride(1), _ptr(hw),
                                        _ptr(start), _ptr(qproj), _ptr(ref), ref.shape[3], _ptr(valid_hw), _ptr(out),
                                        int(out_dtype == torch.float32), B, S, heads, D, L, Q, P, _stream()),
             "mq_msdeform_attn_q_fwd")
    return out


def ml_nms(boxes, labels, nvalid, thresh, max_keep=0, as_bool=True):
    """boxes [B,N,4] fp32 sorted by score desc, labels [B,N] int32, nvalid [B] int32 -> keep [B,N] bool.
    max_keep > 0 with KERNELS["NMS_EARLY_STOP"] (default): the sweep of an image ends once max_keep boxes are kept (mq_ml_nms_topk; the
    max_keep highest-scoring survivors are the same, later boxes read as not kept)."""
    lib = load_library()
    _need_gpu(boxes, labels, nvalid)
    B, N, _ = boxes.shape
    assert boxes.is_contiguous() and labels.is_contiguous() and boxes.dtype == torch.float32
    assert labels.dtype == torch.int32 and nvalid.dtype == torch.int32
    ws = torch.empty(max(lib.mq_ml_nms_workspace_bytes(B, N), 8), dtype=torch.uint8, device=boxes.device)
    keep = torch.empty(B, N, dtype=torch.uint8, device=boxes.device)
    if max_keep > 0 and N <= 6656 and KERNELS["NMS_EARLY_STOP"] == 1:
        _chk(lib.mq_ml_nms_topk(_ptr(boxes), _ptr(labels), _ptr(nvalid), _ptr(ws), _ptr(keep), B, N, float(thresh), int(max_keep), _stream()),
             "mq_ml_nms_topk")
        return keep.bool() if as_bool else keep
    _chk(lib.mq_ml_nms(_ptr(boxes), _ptr(labels), _ptr(nvalid), _ptr(ws), _ptr(keep), B, N, float(thresh), _stream()),
         "mq_ml_nms")
    return keep.bool() if as_bool else keep
