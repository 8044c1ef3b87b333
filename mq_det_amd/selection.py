"""Kernel selection of the MQ-Det HIP path: which of two implementations of an operator runs (defaults, cfg.MODEL.KERNELS, environment MQ_<NAME>),
and the LIVE table the wrappers of mq_det_amd.ops read -- one per host thread.  Split out of ops.py in round 6; ops re-exports
every name, so `ops.KERNELS`, `ops.configure`, `ops.activate`, `ops.KERNEL_DEFAULTS` are these objects."""
import os

# Kernel selection: which of two implementations of an operator runs.  Every variant was A/B'd one by one on the
# MI355X (round 3, GPU call 1, 30 steps each, same box: profiles/r03_call1_switch_ab.txt) and the winners are the defaults below.
# Read ONCE -- `configure()` is called by the detectors' prepare() -- never per call: cfg.MODEL.KERNELS.<NAME> first, then the
# environment variable MQ_<NAME> (A/B runs, tests).
KERNEL_DEFAULTS = {
    "LN_VARIANT": 2,             # 2: mq_layernorm2_fwd (rows in flight, gamma / beta in registers; bit-identical results)      +1.2 %
    "OFFSET_CONV_VARIANT": 3,    # 2: mq_conv3x3_nchw32_v2_fwd (window loads unconditional and in flight; bit-identical to 1)    +4.4 %
                                 # 3: mq_conv3x3_nchw32_group_fwd (all levels of a DyConv layer in ONE launch of persistent workgroups,
                                 #    weights in registers; per-level calls and other shapes: the v2 kernel)   +3.3 % over 2 (r04 call 21)
    "PATCH_MERGE_FUSED": 1,      # 1: mq_patch_merge_ln_fwd (Swin PatchMerging gather + LayerNorm, no pad / cat pass)            +1.6 %
    "FPN_VIA_DCN": 1,            # 1: the three FPN output convs as ONE grouped launch of the fused DCNv2 kernel, zero offsets    +3.8 %
    "NMS_EARLY_STOP": 1,         # 1: mq_ml_nms_topk (the sweep of an image ends once DETECTIONS_PER_IMG boxes are kept)          +0.9 %
    "ATTN_RESIDENT": 1,          # 1: mq_attn_resident_fwd / mq_attn_chunked_fwd (S^T form, keys resident / 256-key chunks)       +4.7 %
    "SWIN_MLP_VARIANT": 2,       # 2: mq_swin_mlp2_fwd (fragment-major weights, 3-deep software pipeline, 14-VALU GELU); 1: the library path (LayerNorm
                                 # kernel + GEMM + GELU + GEMM); the first-generation kernel mq_swin_mlp_fwd is gone (round 5)
    "SWIN_MLP2_FLAGS": -1,       # mq_swin_mlp2_fwd flags: -1 = per width (table GELU everywhere, one pass below C = 384: GPU call 22 of round 6; precise mode: erf, split); else bit 1 = table
                                 # GELU, bit 0 = no tail split, bit 2 = everything through the tail kernel
    "SWIN_QKV_FUSED": 2,         # the Swin qkv projection inside the window attention (mq_window_attn_qkv_fwd): 1 = at C = 96 (0.372 -> 0.137 ms
                                 # per block), 2 = also at C = 192 (weights streamed per head: 0.20 -> 0.137), 0 = GEMM + mq_window_attn_fwd
    "FPN_TOPDOWN_FUSED": 1,      # 1: mq_add_upsample_nearest (lateral += up-sampled coarser level, in place: 25.6 us against 135 us for F.interpolate
                                 # + add on P3 at B = 8, equal outputs on the device -- GPU call 16 of round 3); 0: F.interpolate + add
    "DYRELU_IN_LN": 1,           # 1: the DYReLU of fusion layers 0 .. L-2 is applied by the next layer's LayerNorm (mq_dyrelu_ln_fwd); 0: own pass
    "VLFUSE_I2T_VARIANT": 0,     # mq_vlfuse_i2t_fwd: 0 = Q fragments in registers where they fit (129 .. 160 keys: 0.428 -> 0.370 ms per launch),
                                 # 1 = Q tile in LDS for every caption longer than 128 tokens
    "SWIN_MLP_TAIL_STREAM": 0,   # 1: the tail blocks of mq_swin_mlp2_fwd (those beyond the last full pass of the chip: 52 of 2100 at C = 384, B = 8,
                                 # 64 us in FRONT of the 130 us main kernel) on a side stream BESIDE the main kernel (a tail workgroup fits on a CU next
                                 # to a main one); 0: one after the other.  A/B on the MI355X (GPU call 12): 437 vs 441 images/s over two runs each --
                                 # no gain (the main kernel leaves the tail's waves no issue slots), stays off
    "DYCONV_EPILOGUE_GROUPED": 1,  # 1: mq_dyconv_epilogue_group -- the fuse pass and the DYReLU coefficients of ALL levels of a DyConv layer in two launches
                                 # on the main stream (were 10 launches on five streams behind a fork / join); equal results.  A/B of round 5
                                 # (GPU call 1, 3 alternations x 60 steps): 439.2 / 440.8 / 437.8 against 433.7 / 439.1 / 435.0 images/s: +0.8 %, default
    "BERT_CLAMP_FUSED": 1,       # (round 5, GPU call 1: 0.6 % slower, three of three, and call 11 of round 6 again: off until the end of round 6.  On the final
                                 # step -- GPU call 28 -- a tie: 503.0 against 502.8 images/s over three alternations; on since: 36 torch launches per step
                                 # fewer, the torch-native share of an eager step 9.4 -> see profiles/r06_call29_per_step_summary.txt)
                                 # 1: the +-50000 clamps of the fusion-layer BERT copies inside the kernels around them (mq_clamp_gelu_clamp: clamp -> GELU
                                 # -> clamp in one pass; mq_layernorm_clamp_fwd: clamp of the dense output, LayerNorm, clamp of both outputs): 7 torch
                                 # launches per layer -> 1, equal results; 0: torch.clamp / F.gelu passes
    "PATCH_EMBED_FUSED": 1,      # 1: mq_patch_embed_fwd (Swin PatchEmbed projection + patch_embed.norm + the first norm1 in one pass over the pixels);
                                 # 0: permute copies + library GEMM (K = 48) + two LayerNorm launches (354 us at B = 8)
    "BERT_QKV_FUSED": 1,         # 1: BERT layers = ONE qkv GEMM + mq_attn_text_fwd (V row-major, transposed out of LDS; registers / LDS sized by the
                                 # caption length); 0: q|k GEMM + V^T batched GEMM + mq_attn_resident_fwd (rounds 2-3)
    "FRONT_SIDE_STREAM": 1,      # 1: the image-independent BERT layers on a side stream beside the Swin backbone; 0: on the main stream in front of it (A/B)
    "LANG_SIDE_STREAMS": 0,      # 1: the K / V projections of the second pre-select layer and of GCP blocks 2 .. on side streams beside the serial text chain.
                                 # Measured (round 5, GPU calls 5 / 6): the chain itself is no shorter (1.71 vs 1.68 ms as its own graph) and the WHOLE step
                                 # went from 17.3 to 21 ms -- two more streams than hardware queues (GPU_MAX_HW_QUEUES = 8), branches of the captured
                                 # forward then share queues and serialise.  Off.
    "GCP_ATTN_FUSED": 1,         # mq_gcp_attn_fwd -- the attention half of a GCP block (LayerNorm, to_q, sparse attention, to_out, gate MLP, gated residual,
                                 # next LayerNorm) in one launch.  1: over the measured range -- up to FUSED_TEXT_MAX_ROWS text rows per launch (every workgroup
                                 # streams all 2.1 MB of weights, in MFMA B-fragment order since GPU call 17: 33 us against 154 us for the eight launches at
                                 # B = 8, 102 against 127 us at B = 64; with row-major weights it was 73 / 219 us); 2: always; 0: the eight launches of rounds 2-4
    "BERT_ATTN_QKV_FUSED": 1,    # mq_bert_attn_qkv_fwd -- the q | k | v projection inside the attention launch (one workgroup per (batch item, head); the qkv
                                 # tensor is never written).  1: over the measured range, B x heads <= FUSED_BERT_MAX_WORKGROUPS (weights in MFMA B-fragment
                                 # order since GPU call 17: 23 us against 39 us for the library GEMM + mq_attn_text_fwd at B = 8, 66 against 72 us at B = 64;
                                 # row-major weights: 26 / 82 us); 2: always; 0: never (round 4's pair)
    "POST_FUSED": 1,             # 1: ATSS post-processing as mq_post_select_fwd + mq_post_sort_fwd + mq_ml_nms_topk + mq_post_finalize_fwd (4 launches);
                                 # 0: the round-1..3 chain (5 x torch.topk + box_decode, argsort, gathers, NMS, topk: ~145 launches, 1.1 ms)
    "F32_OPERANDS": 0,           # the SPLIT-PRECISE mode (MODEL.COMPUTE_DTYPE = "float32"; set by configure() from the config, or MQ_F32_OPERANDS): every kernel's
                                 # 16-bit operands are floats -- the *_f32 entry points, the same kernel sources compiled a third time with
                                 # half_t = float and one 16x16x32 MFMA = three fp16 MFMAs on operands split hi + lo (csrc/common.h); library GEMMs run in fp32.
                                 # 1 = on the device: where a kernel's LDS tiles no longer fit the 160 KB of a CU at twice the element size the
                                 # wrappers pick the variant that does (streamed instead of resident operands, smaller tiles) or the plain
                                 # fp32 torch form; 2 = kernel-source emulation with a 320 KB LDS limit (every kernel as in the 16-bit modes)
    "DCN_BDMA": 1,               # DCNv2 / FPN conv weights in LDS-tile order (ops.dcn_weight_tiles, packed once beside the row-major copy) copied
                                 # global -> LDS by LDS-DMA (`buffer_load ... lds`; csrc/dcn_fused.hip BDMA): no weight registers, no ds_write for B.
                                 # 1 (default since GPU call 13 of round 6): every build; 0: never (weights through registers); -1: in the
                                 # split-precise mode only (the default of GPU call 12, when the fp16 variant was 9 % slower: hipcc's wait-count
                                 # pass turned every wait behind the FLAT-encoded copy into vmcnt(0) / lgkmcnt(0)).  Same-box A/B, 3 alternations:
                                 # fp16 0.559 -> 0.515 ms per launch (0.227 -> 0.246 of the MFMA peak), 483 -> 493 images/s; split-precise
                                 # 1.50 -> 1.21 ms, 174 -> 182 images/s (profiles/r06_call13_dcn_bdma_ab.txt)
    "POOLED_TOKENS_FUSED": 1,    # 1: the pooled FPN tokens of the GCP pre-select in one launch (mq_pool2x2_tokens_fwd: bit for bit what five F.avg_pool2d + torch.cat
                                 # return -- torch.equal on the device at the benchmark pyramid and at odd sizes, fp16 and bf16); 0: the torch statement of
                                 # generalized_vl_rcnn_new.py:291-293.  Six launches on the chain between the FPN and the language half -> one: same-box A/B
                                 # (GPU call 31 of round 6, 3 alternations) 500.5 against 496.2 images/s, +0.9 %
    "ALIGN_FUSED": 1,            # 1: mq_align_fused_fwd (heads + alignment + scoring, logits never written); 0: bmm + 5 GEMMs + 5 x mq_align_scores_fwd
}
class _ThreadLocalTable(dict):
    """The LIVE kernel selection, one table per host thread: `activate()` at the top of a forward rewrites the table of the
    thread that runs it, so two models with different selections driven from two host threads cannot change each other's kernels between two
    launches.  A thread that has not activated anything sees the import-time table (defaults <- environment).  Behaves like a dict."""

    def __init__(self, init):
        super().__init__()
        import threading
        object.__setattr__(self, "_tl", threading.local())
        object.__setattr__(self, "_base", dict(init))

    def _cur(self):
        t = self._tl
        if not hasattr(t, "d"):
            t.d = dict(self._base)
        return t.d

    def __getitem__(self, k): return self._cur()[k]
    def __setitem__(self, k, v): self._cur()[k] = v
    def __delitem__(self, k): del self._cur()[k]
    def __contains__(self, k): return k in self._cur()
    def __iter__(self): return iter(self._cur())
    def __len__(self): return len(self._cur())
    def __eq__(self, other): return self._cur() == (other._cur() if isinstance(other, _ThreadLocalTable) else other)
    def __ne__(self, other): return not self.__eq__(other)
    def __repr__(self): return repr(self._cur())
    def get(self, k, default=None): return self._cur().get(k, default)
    def keys(self): return self._cur().keys()
    def values(self): return self._cur().values()
    def items(self): return self._cur().items()
    def clear(self): self._cur().clear()
    def update(self, *a, **k): self._cur().update(*a, **k)
    def copy(self): return dict(self._cur())

    def set_base(self, d):
        """What threads that never activated a selection see (import time / configure() on the main thread)."""
        object.__setattr__(self, "_base", dict(d))


KERNELS = _ThreadLocalTable(KERNEL_DEFAULTS)       # filled from the environment right below configure() (import time), from cfg in prepare()


def configure(cfg=None):
    """(Re)read the kernel selection: defaults <- cfg.MODEL.KERNELS <- environment MQ_<NAME>.  Returns the active table."""
    sel = dict(KERNEL_DEFAULTS)
    node = None
    if cfg is not None:
        node = cfg.MODEL.get("KERNELS", None) if hasattr(cfg.MODEL, "get") else getattr(cfg.MODEL, "KERNELS", None)
    if cfg is not None:
        name = str(cfg.MODEL.get("COMPUTE_DTYPE", "float16") if hasattr(cfg.MODEL, "get") else getattr(cfg.MODEL, "COMPUTE_DTYPE", "float16")).lower()
        sel["F32_OPERANDS"] = 1 if name in ("float32", "fp32", "float") else 0
    for k in sel:
        if node is not None and k in node:
            sel[k] = int(node[k])
        v = os.environ.get("MQ_" + k)
        if v is not None:
            try:
                sel[k] = int(v)
            except ValueError:
                raise ValueError(f"environment variable MQ_{k} = {v!r}: the kernel selection takes integers") from None
    KERNELS.clear()
    KERNELS.update(sel)
    if cfg is None:
        KERNELS.set_base(sel)          # the environment-only table is also what a fresh host thread starts from
    return KERNELS


def activate(sel):
    """Make `sel` (the table a model's prepare() got from configure()) the live selection again.  Every model keeps ITS selection with its
    plan and activates it at the top of each forward, so building a second model with another cfg.MODEL.KERNELS cannot change which
    kernels the first one launches eagerly (its captured HIP graphs hold the selection they were recorded with anyway) -- ADVICE r3."""
    if sel is not None and KERNELS != sel:
        KERNELS.clear()
        KERNELS.update(sel)


configure()        # import time: defaults <- environment MQ_<NAME>, so direct users of ops.* that never call configure() see the env too
