/* mqdet_hip.h -- C ABI of libmqdet_hip.so: the MI355X (gfx950) kernels behind the MQ-Det / GLIP
 * vision-language inference forward.  Plain pointers + sizes, no torch types; all pointers are DEVICE
 * pointers unless stated; every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL =
 * default stream).  Return value: 0 on success, a positive hipError_t for launch failures, a negative
 * value for unsupported shapes / arguments.  The caller owns and allocates every buffer (outputs and
 * workspaces); nothing is retained between calls.
 *
 * Each entry point names the reference interface it replaces (paths under the reference repository
 * YifanXu74/MQ-Det).  The Python-side binding is mq_det_amd/ops.py (ctypes); the binding a maintainer of
 * the reference would add is shown in INTEGRATION.md.
 */
#ifndef MQDET_HIP_H
#define MQDET_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* Library / ABI version (bumped on any signature change). */
int mq_abi_version(void);

/* Fused multi-head attention forward  O = softmax(clamp(scale*Q.K^T, +-clamp) + key_bias) V,
 * fp16 in/out, fp32 accumulate; logits never leave the chip.  D in {32, 64}.
 *   q, k: element (b,i,h,d) at base + b*bs + i*rs + h*hs + d (batch / row / head strides in elements; a head
 *   stride of 0 shares the operand across heads -- used by the folded VLFuse projections),
 *   vt = V transposed: element (b,h,d,j) at vt + b*vt_bs + h*vt_hs + d*vt_rs + j (strides % 8 == 0, columns
 *   Nk..ceil8(Nk) must be finite), o [B,Nq,H*D] at o + b*o_bs + i*o_rs + h*D + d; key_bias fp32 (b,h,j) at
 *   key_bias + b*bias_bs + h*bias_hs + j or NULL; clamp <= 0 disables.  kv_len [B] int32 or NULL: keys >= kv_len[b]
 *   are masked by the caller's key_bias (text padding) and their tiles are skipped (exactly-zero contribution).
 *   qk_mask uint8 or NULL: per-(query, key) mask, element (b,h,i,j) at qk_mask + b*mask_bs + h*mask_hs + i*mask_rs + j,
 *   1 = key j is hidden from query i (strides in bytes, 0 shares the mask across batch / heads): the sub-sentence block mask
 *   of MQ-GroundingDINO's BERT (bertwarper.py:273-320 -> :140-144) and the `attn_mask` of its text-enhancer layers
 *   (transformer_vanilla.py:108-113).
 *   nsplit > 1 splits the key range over grid.z (few queries / many keys) and needs
 *   mq_attn_workspace_bytes() bytes of workspace.
 * Replaces the unfused bmm -> (+mask) -> softmax -> bmm chains of
 *   maskrcnn_benchmark/modeling/rpn/modeling_bert.py:119-170 (BertSelfAttention, with clamp) and HF BertSelfAttention,
 *   maskrcnn_benchmark/modeling/language_backbone/modeling_bert_new.py:204-240 (MaskedCrossAttention, dense), and the
 *   nn.MultiheadAttention calls of MQ-GroundingDINO (decoder self-attention 8 x 32 over 900 queries and text cross-attention,
 *   groundingdino_new/models/GroundingDINO/transformer.py:895-911; text enhancer 4 x 64, transformer_vanilla.py:92-123).
 *   (The 8 x 256 VLFuse attention has its own kernels, mq_vlfuse_*.) */
long mq_attn_workspace_bytes(int B, int H, int Nq, int D, int nsplit);
int mq_attn_fwd(const void* q, const void* k, const void* vt, void* o, const float* key_bias, const int* kv_len,
                const unsigned char* qk_mask, long mask_bs, long mask_hs, long mask_rs, void* workspace,
                int B, int H, int Nq, int Nk, int D,
                long q_bs, long q_rs, long q_hs, long k_bs, long k_rs, long k_hs, long vt_bs, long vt_rs, long vt_hs,
                long o_bs, long o_rs, long bias_bs, long bias_hs, float scale, float clamp, int nsplit, void* stream);

/* The same operator for SHORT key sequences (Nk <= 256: every attention over the text tokens): all keys of a (batch, head) resident
 * in LDS, S^T = K Q^T so that P stays in registers, exact two-pass softmax (csrc/attn_resident.hip).  Arguments as mq_attn_fwd
 * without the key split; additionally o_rs % 4 == 0 and, with a qk_mask, Nk % 4 == 0, mask strides % 4 == 0 and a 4-byte aligned
 * mask base (the mask is read one 32-bit word = 4 keys at a time; -3 otherwise).  Returns -1 for Nk > 256 or D not in {32, 64}.  Default for these shapes since
 * round 3 (KERNELS["ATTN_RESIDENT"] = 1: +4.7 % end to end on its own, profiles/r03_call1_switch_ab.txt). */
int mq_attn_resident_fwd(const void* q, const void* k, const void* vt, void* o, const float* key_bias, const int* kv_len,
                         const unsigned char* qk_mask, long mask_bs, long mask_hs, long mask_rs, int B, int H, int Nq, int Nk, int D,
                         long q_bs, long q_rs, long q_hs, long k_bs, long k_rs, long k_hs, long vt_bs, long vt_rs, long vt_hs,
                         long o_bs, long o_rs, long bias_bs, long bias_hs, float scale, float clamp, void* stream);

/* ... and for LONG key sequences without a per-(query, key) mask (GCP pre-select, MQ-GroundingDINO decoder self-attention): the
 * same S^T formulation over chunks of 256 keys with a running (max, sum, O) rescaled once per chunk, register prefetch of the next
 * chunk, one barrier per chunk; key split like mq_attn_fwd (same workspace size, mq_attn_workspace_bytes).  Arguments as mq_attn_fwd
 * without the qk_mask; o_rs % 4 == 0.  Selected together with mq_attn_resident_fwd (KERNELS["ATTN_RESIDENT"]). */
int mq_attn_chunked_fwd(const void* q, const void* k, const void* vt, void* o, const float* key_bias, const int* kv_len,
                        void* workspace, int B, int H, int Nq, int Nk, int D,
                        long q_bs, long q_rs, long q_hs, long k_bs, long k_rs, long k_hs, long vt_bs, long vt_rs, long vt_hs,
                        long o_bs, long o_rs, long bias_bs, long bias_hs, float scale, float clamp, int nsplit, void* stream);

/* Swin (shifted-)window attention with pad / roll / window partition folded into addressing.
 *   qkv [B,H,W,3C] fp16, qkv_bias [3C] fp16 (pad tokens), rel_bias [heads,NP,NP] fp32 (rows = query, cols = key,
 *   zero-padded from N = ws*ws to NP = 64 when N <= 64 -- window 7 -- or 160 when N <= 160 -- Swin-L, window 12),
 *   out [B,H,W,C] fp16; C == heads*32.
 * Replaces maskrcnn_benchmark/modeling/backbone/swint.py:111-142 (WindowAttention.forward) together with the
 *   pad/roll/window_partition/window_reverse/crop copies of SwinTransformerBlock.forward (:201-234) and the
 *   per-forward SW-MSA mask construction of BasicLayer.forward (:354-373). */
int mq_window_attn_fwd(const void* qkv, const void* qkv_bias, const float* rel_bias, void* out,
                       int B, int H, int W, int C, int heads, int ws, int shift, void* stream);
/* The same operator with the qkv projection INSIDE (csrc/window_attn.hip, round 3): x [B,H,W,C] fp16 = norm1(x) on the unpadded tokens,
 * w [3C, C] / bias [3C] fp16 = attn.qkv (nn.Linear layout; swint.py:97,111-117), rel_bias / out as above -- the [B,H,W,3C] qkv tensor
 * (310 MB per block at stage 1, B = 8) is never written.  One wave per window; Q^T / K^T / V come out of the projection MFMAs in the
 * fragment layouts of the attention MFMAs.  C = 96 (heads = 3), windows of at most 64 tokens; -1 otherwise. */
int mq_window_attn_qkv_fwd(const void* x, const void* w, const void* bias, const float* rel_bias, void* out,
                           int B, int H, int W, int C, int heads, int ws, int shift, void* stream);

/* GCP sparse cross-attention: text token t attends to the vision rows idx[b,t,0..S) (-1 = none).
 *   q [B,T,512] fp16, kv [B,V,1024] fp16 (k|v of the UNIQUE vision tokens), idx [B,T,S] int32 (any S >= 0), out [B,T,512].
 * Replaces MaskedCrossAttention.forward (sparse branch) + _construct_sparse_inputs,
 *   maskrcnn_benchmark/modeling/language_backbone/modeling_bert_new.py:162-184,204-240. */
int mq_gcp_sparse_attn_fwd(const void* q, const void* kv, const int* idx, void* out, int B, int T, int V, int S,
                           int heads, int dim_head, void* stream);

/* GCP conditional gate fused into the residual:  out = sup * tanh(w2 . gelu(h)) + x   (rows M, widths C / G).
 *   sup, h, w2 fp16; x and out are the residual stream: fp32 when x_f32 != 0, else fp16.
 *   gate_out [M] fp32 optional (VISION_QUERY.RETURN_ATTN_GATE_VALUE).
 * Replaces GatedCrossAttentionBlock.forward lines modeling_bert_new.py:359,368. */
int mq_gcp_gate_residual_fwd(const void* sup, const void* h, const void* w2, const void* x, int x_f32, void* out,
                             float* gate_out, long M, int C, int G, void* stream);

/* VLFuse image side in ONE launch (heads <= 8, head dim 256, text tokens T <= 256), projections folded into the text operands:
 *   out[b,n,:] = v_ln[b,n,:] + out_bias + sum_h softmax_t( clamp(v_ln[b,n,:] . kf[b,h,t,:] + bias[b,h,t], +-clamp) ) vo[b,h,t,:]
 *   v_ln [B,N,256] fp16 (LN(v): queries AND residual), kf / vo [B,heads,T,256] fp16 (folded text keys / values),
 *   bias [B,heads,T] fp32 or NULL (<= -1e29: key masked), kv_len [B] int32 or NULL (keys >= kv_len[b] masked),
 *   max_kv: host-side upper bound of kv_len (<= 0: T) -- the caller guarantees kv_len[b] <= max_kv,
 *   out_bias [256] fp16, out [B,N,256] fp16.
 * Replaces BiMultiHeadAttention's image branch: v_proj, the [B*8, N, T] logits, clamp, softmax over text,
 *   bmm with values_l, out_v_proj, and the gamma_v residual of BiAttentionBlock
 *   (maskrcnn_benchmark/utils/fuse_helper.py:221-279,290-300,424; heads = 8), and the same branch of MQ-GroundingDINO's
 *   feature-enhancer fusion (groundingdino_new/models/GroundingDINO/fuse_modules.py:146-249,286-296; heads = 4). */
int mq_vlfuse_i2t_fwd(const void* v_ln, const void* kf, const void* vo, long kv_bs, long kv_hs, long kv_ts, const float* bias, const int* kv_len,
                      const void* out_bias, void* out, int B, int N, int T, int heads, int max_kv, float clamp, int variant,
                      void* stream);
/* kv_bs / kv_hs / kv_ts (ABI 30): element strides of kf AND vo over (batch item, head, text token) -- element (b, h, t, :) is read at
 *   b * kv_bs + h * kv_hs + t * kv_ts, 256 consecutive elements; all three 0: a contiguous [B,heads,T,256] tensor.  The caller of the
 *   fusion layer passes views of the ONE projection GEMM's output [B, T, heads*256 | heads*256 | ...] (kv_hs = 256, kv_ts = its row
 *   length): no `permute().contiguous()` copies of the folded keys / values.  Strides and both pointers must be multiples of 16 bytes. */
/* variant: 0 = default; 1 = Q tile in LDS for every caption longer than 128 tokens (A/B: by default 129 .. 160 keys keep the Q fragments
 *   in registers); >= 100: ablation timings of tools/microbench.py (results undefined). */

/* VLFuse text side: keys = values = image tokens, split over the keys (nsplit >= 1) + merge:
 *   out[b,t,h*256:(h+1)*256] = sum_n softmax_n( clamp(kf[b,h,t,:] . v_ln[b,n,:], +-clamp) ) v_ln[b,n,:]
 *   kf [B,heads,T,256]; workspace: mq_vlfuse_t2i_workspace_bytes(B, T, nsplit) bytes of device memory (sized for 8 heads);
 *   out [B,T,heads*256] fp16;
 *   key_mask uint8 [B, key_mask_bs] or NULL: 1 = image token n is padding and masked as a key (`attention_mask_v`,
 *   fuse_modules.py:205-210: images of different sizes in one padded batch); key_mask_bs % 4 == 0 and >= 64*ceil(N/64);
 *   kv_len [B] int32 or NULL: caption length -- 16-row blocks that hold only padding tokens (rows >= kv_len[b]) are not
 *   computed and come back as zeros (padding rows never influence a detection: masked as keys, never scored).
 * Replaces the text branch of BiMultiHeadAttention: the transposed logits, their softmax over image tokens and the
 *   bmm with values_v (fuse_helper.py:246-262,281-288); values_v_proj / out_l_proj are applied to the result by the
 *   caller as one folded [768, 2048] weight. */
long mq_vlfuse_t2i_workspace_bytes(int B, int T, int nsplit);
int mq_vlfuse_t2i_fwd(const void* kf, long kv_bs, long kv_hs, long kv_ts, const void* v_ln, const int* kv_len, const unsigned char* key_mask,
                      long key_mask_bs, void* workspace, void* out, int B, int N, int T, int heads, int nsplit, int max_kv, float clamp, int variant,
                      void* stream);
/* kv_bs / kv_hs / kv_ts: element strides of kf as for mq_vlfuse_i2t_fwd (0, 0, 0: contiguous). */
/* max_kv: host-side upper bound of kv_len (0 = T): sizes the grid -- the live (head, 16-row block) units of an image are packed
 *   densely over the waves of its workgroups.  variant: 0 (values >= 100: ablation timings of tools/microbench.py, results undefined). */

/* Row LayerNorm with the residual add fused in, mixed-precision streams, fp32 statistics; C % 8 == 0, C <= 3072.
 *   s = x (+ res);  x is fp32 when x_f32 != 0 else fp16, res likewise (res_f32); res may be NULL.
 *   y    [rows,C] fp16 = (s - mean) * rstd * gamma + beta    (may be NULL)
 *   y32  [rows,C] fp32 = the same, unrounded                 (may be NULL; post-LN residual stream of the BERT layers)
 *   xsum [rows,C]      = s (may be NULL; needs res): fp32 when x or res is fp32, otherwise fp16 -- s is then rounded to
 *                        fp16 before the statistics, exactly what a separate elementwise add would hand to LayerNorm.
 * Replaces every nn.LayerNorm call on the path (backbone/swint.py:198,240,281,425,611; HF BertLayer / embeddings;
 *   language_backbone/modeling_bert_new.py:121,150-153; utils/fuse_helper.py:420-421) together with the residual adds
 *   in front of them (swint.py:236,240; BertSelfOutput / BertOutput). */
int mq_layernorm_fwd(const void* x, int x_f32, const void* res, int res_f32, const void* gamma, const void* beta, void* y,
                     float* y32, void* xsum, long rows, int C, float eps, void* stream);

/* The same operator, arguments and results (bit-identical: the summation order is kept) with a different load schedule: chunk count
 * per lane fixed at compile time, gamma / beta in registers for the whole block, up to 4 rows per lane group in flight, every load of
 * an iteration issued before the first is consumed (csrc/layernorm2.hip).  Default since round 3 (KERNELS["LN_VARIANT"] = 2);
 * bit for bit the results of mq_layernorm_fwd (device and tests/simt). */
int mq_layernorm2_fwd(const void* x, int x_f32, const void* res, int res_f32, const void* gamma, const void* beta, void* y,
                     float* y32, void* xsum, long rows, int C, float eps, void* stream);
/* mq_layernorm_fwd with the three clamps of the VLDyHead BERT copies inside (rpn/modeling_bert.py:242-272: the output of the dense layer,
 * then the LayerNorm output): s = med3(x, +-clamp) (+ res), y = clamp(round16(LN(s))), y32 = clamp(LN(s)); equal to clamp -> mq_layernorm_fwd
 * -> clamp, clamp bit for bit.  clamp > 0.  KERNELS["BERT_CLAMP_FUSED"]. */
int mq_layernorm_clamp_fwd(const void* x, int x_f32, const void* res, int res_f32, const void* gamma, const void* beta, void* y, float* y32,
                           void* xsum, long rows, int C, float eps, float clamp, void* stream);
/* out = clamp(gelu(clamp(x))) elementwise on n 16-bit values (n % 8 == 0), exact (erf) GELU in fp32 rounded once: the intermediate
 * activation of the clamped BERT copies (rpn/modeling_bert.py:255-259) in one pass instead of torch's three. */
int mq_clamp_gelu_clamp(const void* x, void* out, long n, float clamp, void* stream);

/* Swin PatchMerging up to its LayerNorm in one kernel: y[b,i,j,:] = LayerNorm_{4C}(concat(x[b,2i,2j], x[b,2i+1,2j], x[b,2i,2j+1],
 *   x[b,2i+1,2j+1])), zero beyond an odd H / W.  x [B,H,W,C] fp16 or fp32 (x_f32), contiguous; gamma / beta [4C] fp16; y fp16
 *   [B, ceil(H/2)*ceil(W/2), 4C]; C % 8 == 0, 4C <= 3072.  Bit-identical to F.pad + torch.cat + mq_layernorm_fwd.
 * Replaces maskrcnn_benchmark/modeling/backbone/swint.py:264-281 (pad, four strided slices, cat, norm).  Default since round 3
 * (KERNELS["PATCH_MERGE_FUSED"] = 1). */
int mq_patch_merge_ln_fwd(const void* x, int x_f32, const void* gamma, const void* beta, void* y, int B, int H, int W, int C, float eps,
                          void* stream);

/* MLP half of a Swin block in one kernel (LayerNorm prologue, fc1, exact GELU, fc2, residual; the 4C-wide hidden activation stays in
 * registers), C in {96, 192, 384}:
 *   x' = x + delta;  out = x' + fc2(gelu(fc1(LN(x'; ln_g, ln_b, eps))));  y = LN(out; next_g, next_b, eps_next) (optional)
 *   x [M,C] fp32 (residual stream), delta [M,C] fp16 or NULL, b1 [4C], b2 [C] fp16, out [M,C] fp32, y [M,C] fp16 or NULL.
 *   Returns -1 for other C (callers fall back to library GEMMs).
 * Replaces SwinTransformerBlock's  x = x + drop_path(mlp(norm2(x)))  (backbone/swint.py:238-240, Mlp :13-31) and, through `y`, the
 *   following block's norm1 (:198) or the stage's output norm (:611).
 * (csrc/swin_mlp2.hip -- the second generation; the first, mq_swin_mlp_fwd on row-major weights, was removed in round 5.)  The weights
 * arrive FRAGMENT-MAJOR so that staging is a linear LDS-DMA copy and every LDS read is conflict-free;
 * GELU(chunk j), fc1(chunk j + 1) and fc2(chunk j - 1) are issued together (software pipeline).  A launch lasts a whole number of
 * passes of the chip (one 16-token block per wave); when only a few blocks remain beyond the last full pass they go to a second
 * kernel that splits the hidden dimension of ONE block over the waves of a workgroup (1 / 8 of a pass instead of a whole one).
 * flags bit 1: GELU through a 768-entry interpolation table of Phi (|error| < 8e-6) instead of the 14-instruction erf formula;
 * bit 0: no pass / tail split; bit 2: every block through the tail kernel (tests).
 *   w1f [(4C/32 + 2) * (C/16) * 512] fp16: block (chunk j, hb in {0,1}, ks) holds for lane l  fc1.weight[32j + 16hb + (l & 15)][32ks + 8(l >> 4) .. +7];
 *        two all-zero chunks follow the last one (the pipeline reads two chunks ahead);
 *   w2f [(4C/32) * (C/16) * 512] fp16: block (chunk j, ct) holds for lane l  w2p[16ct + (l & 15)][32j + 8(l >> 4) .. +7], w2p = fc2.weight with
 *        the k-slots of every 32-block permuted (slot 8g+t <- hidden 4g+t for t < 4, 16+4g+t-4 for t >= 4: the transposed GEMM chain hands
 *        its accumulators to the second MFMA as B fragments).   (Python: ops.swin_mlp2_pack(fc1.weight, fc2.weight).) */
int mq_swin_mlp2_fwd(const float* x, const void* delta, const void* ln_g, const void* ln_b, float eps, const void* w1f,
                     const void* b1, const void* w2f, const void* b2, float* out, const void* next_g, const void* next_b,
                     float eps_next, void* y, long M, int C, int flags, void* stream);

/* 3x3 convolution (pad 1, stride 1|2) and DCNv2 (modulated deformable 3x3) as one implicit-GEMM MFMA kernel,
 * NHWC fp16, fp32 accumulation over the whole K = 9*C in a fixed order (bitwise reproducible).
 *   x [B,H,W,C] (batch stride x_bs elements, C % 32 == 0), w [Npad, 9*C] fp16 with k = tap*C + c (Npad = 32 if
 *   N <= 32 else 256; rows >= N zero), bias [N] fp16 or NULL, out [B*Ho*Wo, out_ld] fp16 (first N columns written).
 *   mq_dcnv2_fwd additionally takes om [B,27,oH,oW] fp32 NCHW = 18 offsets + 9 mask LOGITS (flags bit 0 set: mask PROBABILITIES, the
 *   argument of the reference operator -- no sigmoid is applied then), indexed flat by the output
 *   dims like the reference kernel (the buffer may come from another pyramid level); N must be 256.  With stats != NULL it
 *   also emits the GroupNorm / scale-attention statistics of its output (layout of mq_dyconv_stats with
 *   mq_dcnv2_stats_blocks(H, W, stride) blocks per image -- one per tile of 128 output positions: ceil(Ho * Wo / 128) since round 6 (tiles are
 *   consecutive positions in band order; with MQ_DCN_RASTER=0 in the environment: ceil(Ho / 8) * ceil(Wo / 16) patches as before); wy [Ho],
 *   wx [Wo] position weights or NULL for 1/(Ho*Wo)).
 * mq_dcnv2_fwd replaces _C.modulated_deform_conv_forward (maskrcnn_benchmark/csrc/vision.cpp:11-12,
 *   csrc/cuda/deform_conv_cuda.cu:496-575, deform_conv_kernel_cuda.cu:578-640) without the im2col buffer;
 * mq_conv3x3_fwd replaces the nn.Conv2d 3x3 calls of backbone/fpn.py:41,141-146 and the DyConv offset conv
 *   (rpn/vldyhead.py:186,214). */
int mq_conv3x3_fwd(const void* x, const void* w, const void* bias, void* out, int B, int H, int W, int C, long x_bs,
                   int N, int out_ld, int stride, void* stream);
/* 3x3 convolution (pad 1, stride 1) with N <= 32 output channels, fp32 NCHW output: the 27-channel offset / mask conv of
 * every DyConv level.  x [B,H,W,C] fp16 NHWC (batch stride x_bs; C % 32 == 0, C <= 256), w [32, 9*C] fp16 (k = tap*C + c,
 * rows >= N zero), bias [N] fp16 or NULL -> out [B,N,H,W] fp32.  The input window of an 8x16 output patch is staged once
 * in LDS.  Replaces nn.Conv2d(256, 27, 3) + the fp32 offset/mask split of rpn/vldyhead.py:186,214-216. */
int mq_conv3x3_nchw32_fwd(const void* x, const void* w, const void* bias, float* out, int B, int H, int W, int C, long x_bs,
                          int N, void* stream);

/* The same operator, arguments and results (bit-identical) with an unconditional, fully in-flight load schedule for the input window
 * and the weight prefetch (csrc/conv_small2.hip).  Default since round 3 (KERNELS["OFFSET_CONV_VARIANT"] = 2); equal outputs. */
int mq_conv3x3_nchw32_v2_fwd(const void* x, const void* w, const void* bias, float* out, int B, int H, int W, int C, long x_bs,
                          int N, void* stream);
/* The offset / mask conv of ONE DyConv layer for all pyramid levels in one launch (csrc/conv_small3.hip; rpn/vldyhead.py:205-215 applies the
 * same nn.Conv2d(256, 27, 3) to every level).  `levels` is a HOST array of <= 8 entries copied into the kernel arguments: x [B,H,W,256]
 * 16-bit NHWC (batch stride x_bs elements), out [B,N,H,W] fp32 NCHW.  w [32, 9*256] / bias [N] as mq_conv3x3_nchw32_fwd.  Persistent
 * workgroups (one per CU) hold the weights in registers and walk the tiles of all levels; the contraction is split over the 8 waves by
 * channel and summed in a fixed order, so results equal mq_conv3x3_nchw32_fwd's to fp32 rounding (not bit for bit).
 * Returns -1 for C != 256, N > 32, more than 8 levels (callers then use mq_conv3x3_nchw32_v2_fwd per level).
 * KERNELS["OFFSET_CONV_VARIANT"] = 3. */
typedef struct mq_conv_level {
  const void* x; float* out; long x_bs; int H, W;
} mq_conv_level;
int mq_conv3x3_nchw32_group_fwd(const mq_conv_level* levels, int nl, const void* w, const void* bias, int B, int C, int N, void* stream);
/* Pooled FPN tokens the GCP pre-select attends to (generalized_vl_rcnn_new.py:291-293: `torch.cat([F.avg_pool2d(f, 2) ... tokens], 1)`) in ONE
 * launch (ABI 31): levels[i].x = NHWC 16-bit [B,H,W,C] (batch stride x_bs elements; `out` unused), out [B, sum_l (H_l/2)*(W_l/2), C] 16-bit:
 * token (l, y, x) = mean of the 2 x 2 window (floor sizes), fp32 sum in row-major window order, one rounding -- what ATen's NHWC pool returns. */
int mq_pool2x2_tokens_fwd(const mq_conv_level* levels, int nl, void* out, int B, int C, void* stream);
int mq_dcnv2_stats_blocks(int H, int W, int stride);
/* One launch for up to 16 DCNv2 calls (the 13 branches of one DyConv layer): `branches` is a HOST array, copied into the
 * kernel arguments; fields as the arguments of mq_dcnv2_fwd.  The tiles of all branches form one work list, so small
 * pyramid levels do not pay a launch (and a partial wave of workgroups) each. */
typedef struct mq_dcn_branch {
  const void* x; const float* om; const void* w; const void* bias; void* out; float* stats; const float* wy; const float* wx;
  long x_bs;
  int B, H, W, C, oH, oW, N, out_ld, stride, flags;          /* flags bit 0: om[:, 18:27] are probabilities, not logits; bit 1: the caller
                                                                 promises all-zero offsets and mask 1 (a plain 3 x 3 conv: when every branch of a
                                                                 launch says so only one corner per tap is gathered -- same results);
                                                                 bit 2 (round 6, on every branch of a launch or on none: -5 otherwise): `w` is in
                                                                 LDS-TILE ORDER -- per k-step ks = (channel slice) * 9 + tap one 32 KB block that
                                                                 is the byte image of the kernel's B tile: 256 rows x 64 channels with the 16-byte
                                                                 chunk c of row r at position c ^ (r & 7) (*_f32: [hi | lo] fp16 planes of 256
                                                                 rows x 32 channels) -- and is copied global -> LDS by LDS-DMA
                                                                 (mq_det_amd.ops.dcn_weight_tiles; same results) */
} mq_dcn_branch;
int mq_dcnv2_group_fwd(const mq_dcn_branch* branches, int n, void* stream);
int mq_dcnv2_fwd(const void* x, const float* om, const void* w, const void* bias, void* out, float* stats, const float* wy,
                 const float* wx, int B, int H, int W, int C, long x_bs, int oH, int oW, int N, int out_ld, int stride,
                 int flags, void* stream);

/* DyConv epilogue (GroupNorm(16) + bilinear up-sampling of the level+1 branch + scale attention + branch mean,
 * then DYReLU), NHWC fp16 with fp32 statistics; C == 256.
 *   mq_dyconv_stats : y [B,n,C] -> sums [B,ceil(n/256),C,3] fp32 per-block partials (sum, sum sq, weighted sum; fixed
 *                     reduction order = reproducible); wy [n/W], wx [W] fp32 give the per-pixel weights wy*wx
 *                     (spatial mean of the up-sampled map) or NULL for 1/n.
 *   mq_dyconv_coef  : sums [B,nblk,C,3] (nblk <= 0: ceil(n/256), the mq_dyconv_stats layout; mq_dcnv2_fwd's fused
 *                     statistics use nblk = mq_dcnv2_stats_blocks) + GN gamma/beta fp16 [C] + AttnConv weight [C] / bias [1] fp32 -> coef [B,C,2] fp32
 *                     (a*rstd*gamma, a*(beta - mean*rstd*gamma)), a = h_sigmoid(relu(w . pooled + b)) / nbranches.
 *   mq_dyconv_fuse  : out [B,H*W,C] (batch stride out_bs elements: a level's slice of the [B,N,C] pyramid token buffer)
 *                     = sum_k coef_k[.,0]*y_k^ + coef_k[.,1], y_k^ = y_k or its bilinear
 *                     (align_corners) sample from (hs_k, ws_k); pool [B,ceil(H*W/128),C] fp32 per-block sums of out.
 *   mq_dyrelu_coef  : pool, fc.0 / fc.2 weights+biases fp16 -> coef [B,4,C] fp32 (a1,b1,a2,b2).
 *   mq_dyrelu_apply : x [B,n,C] (batch stride x_bs) <- max(a1 x + b1, a2 x + b2) in place.
 * Replaces DyConv.forward's post-conv part, maskrcnn_benchmark/modeling/rpn/vldyhead.py:148-152,224-242 and
 *   DYReLU.forward, maskrcnn_benchmark/layers/dyrelu.py:78-112. */
int mq_dyconv_stats(const void* y, float* sums, const float* wy, const float* wx, int B, int n, int W, int C, void* stream);
/* mq_dyconv_coef for all branches of a DyConv layer in one launch (`branches` is a HOST array of <= 16 entries; sums of
 * branch i have nblk_i blocks and n_i positions; nbranches_i = number of branches fused into that branch's level). */
typedef struct mq_coef_branch {
  const float* sums; const void* gamma; const void* beta; float* coef; int nblk, n, nbranches, reserved;
} mq_coef_branch;
int mq_dyconv_coef_group(const mq_coef_branch* branches, int nbr, const float* attn_w, const float* attn_b, int B, int C, int G,
                         float eps, void* stream);
int mq_dyconv_coef(const float* sums, const void* gamma, const void* beta, const float* attn_w, const float* attn_b,
                   float* coef, int B, int n, int nblk, int C, int G, float eps, int nbranches, void* stream);
int mq_dyconv_fuse(const void* y0, const float* coef0, int hs0, int ws0, const void* y1, const float* coef1, int hs1,
                   int ws1, const void* y2, const float* coef2, int hs2, int ws2, int nbranches, void* out, long out_bs,
                   float* pool, int B, int H, int W, int C, void* stream);
int mq_dyrelu_coef(const float* pool, const void* w0, const void* b0, const void* w2, const void* b2, float* coef,
                   int B, int n, int C, void* stream);
/* The epilogue of a DyConv layer for ALL pyramid levels in two launches (`levels` is a HOST array of <= 8 entries copied into the kernel
 * arguments): mq_dyconv_fuse of every level as one work list, then mq_dyrelu_coef of every level (grid B x levels).  Per level: the
 * arguments of those two entry points -- y / coef / hs / ws of its 1 .. 3 branches, out (batch stride out_bs), pool [B, ceil(H*W/128), C]
 * fp32 workspace, relu_coef [B,4,C] fp32 out.  Same arithmetic and summation order as the per-level entry points: equal results.
 * C == 256.  KERNELS["DYCONV_EPILOGUE_GROUPED"]. */
typedef struct mq_fuse_level {
  const void* y[3]; const float* coef[3]; int hs[3], ws[3];
  int nbranches, H, W, reserved;
  void* out; long out_bs; float* pool; float* relu_coef;
} mq_fuse_level;
int mq_dyconv_epilogue_group(const mq_fuse_level* levels, int nl, const void* w0, const void* b0, const void* w2, const void* b2,
                             int B, int C, void* stream);
int mq_dyrelu_apply(void* x, const float* coef, int B, int n, int C, long x_bs, void* stream);
/* FPN top-down step (backbone/fpn.py:82-95) in place: dst [B,H,W,C] += nearest-up-sampled src [B,Hc,Wc,C] (NHWC 16-bit, C % 8 == 0;
 * source index = min(floor(i * (float)(in / out)), in - 1) as F.interpolate(mode="nearest", size=(H, W))). */
int mq_add_upsample_nearest(void* dst, const void* src, int B, int H, int W, int Hc, int Wc, int C, void* stream);
/* y = LayerNorm(DYReLU(x)) on the head's pyramid token buffer: the DYReLU of a DyConv layer (vldyhead.py:160-188, 245) applied by the
 * ONLY reader of that layer's output, layer_norm_v of the next fusion layer (fuse_helper.py:398).  x, y [B,N,256] 16-bit (x: batch
 * stride x_bs elements), coef [NL][B][4][256] fp32 = mq_dyrelu_coef's output per pyramid level, row_first: NL + 1 HOST ints (level l =
 * rows [row_first[l], row_first[l+1]) of every image; NL <= 8).  DYReLU's result is consumed in fp32. */
int mq_dyrelu_ln_fwd(const void* x, long x_bs, const float* coef, const int* row_first, int NL, const void* gamma, const void* beta,
                     float eps, void* y, int B, int N, int C, void* stream);

/* Region-word alignment scores for the L labels of the caption.
 *   dot [B,HW,T] fp16 (fp32 when dot_f32 != 0; batch stride dot_bs elements, <= 0: HW*T), tbias [B,T] fp32,
 *   tokidx [L,MT] int32 token positions per label, -1 padded (batch stride tok_bs elements: 0 = one caption for the whole
 *   batch, L*MT = one caption per batch item), ctr [B,HW] fp16 -> out [B,HW,L] fp32
 *   ((cls > thr) ? max(cls*sigmoid(ctr), FLT_MIN) : -1), cls_out [B,HW,L] fp32 optional.  Token -> class aggregation
 *   (DYHEAD.SCORE_AGG): agg 0 = MEAN (the reference default), 1 = MAX, 2 = POWER (prod^(1/n)); ONEHOT is expressed by the
 *   host as MEAN over a one-token index (column j <- token j).  A label without tokens scores 0.
 * Replaces vldyhead.py:884-887 (bias, clamp) + rpn/inference.py:656-683,772-824. */
int mq_align_scores_fwd(const void* dot, int dot_f32, const float* tbias, const int* tokidx, long tok_bs, const void* ctr,
                        float* out, float* cls_out, int B, int HW, int T, int L, int MT, float thr, long dot_bs, int agg,
                        void* stream);

/* Prediction heads + alignment + per-location scoring in ONE kernel (csrc/align_fused.hip): bbox_pred / centerness 1x1 convs with the
 * per-level Scale, dot-product region-word logits (+ bias, clamp), sigmoid, token -> class aggregation, threshold, x sigmoid(centerness).
 * The [B, N, T] logits are never written (logits != NULL: fp32 dot products WITHOUT the bias, for parity tests).
 *   tok [B,N,256] fp16 (all pyramid levels concatenated), tk [B,T,256] fp16 = projected text tokens / exp(log_scale), tbias [B,T] fp32,
 *   wbc [16,256] fp16 (rows 0-3 bbox_pred.weight, 4 centerness.weight, 5-15 zero), bbc [8] fp32 biases, scales [NL] fp32,
 *   tokidx [L,MT] (tok_bs = 0) or [B,L,MT] (tok_bs = L*MT) int32 (-1 padded), lvl_off [NL+1] HOST ints (token offset of every level, last = N),
 *   kv_max = upper bound of the live text tokens (0: T); agg 0 MEAN / 1 MAX / 2 POWER.
 *   Outputs are LEVEL-MAJOR (level l contiguous at element offset lvl_off[l]*B*width): ranked / cls_out (or NULL) [B,HW_l,L] fp32,
 *   reg [B,HW_l,4] fp32; ctr_out [B,N] fp32 centerness logits.
 * Replaces VLDyHead.forward's head part (rpn/vldyhead.py:853-888) + ATSSPostProcessor.forward_for_single_feature_map's scoring
 * (rpn/inference.py:656-683, convert_grounding_to_od_logits[_v2] :772-824) for all levels. */
int mq_align_fused_fwd(const void* tok, const void* tk, const float* tbias, const void* wbc, const float* bbc, const float* scales,
                       const int* tokidx, long tok_bs, const int* lvl_off, float* ranked, float* cls_out, float* reg, float* ctr_out,
                       float* logits, int B, int N, int T, int kv_max, int L, int MT, int NL, float thr, int agg, void* stream);

/* Decode + clip the top-K candidates of one level into the per-image candidate arrays (at column out_off).
 *   val/flat [B,K] (score, flat index loc*L + l), reg [B,HW,4] fp16 or (reg_f32) fp32, anchors [HW,4] fp32, label_ids [L] int32 (batch
 *   stride lab_bs elements, 0 = shared), im_wh [B,2] fp32 (w,h) -> boxes [B,out_stride,4] fp32, scores [B,out_stride]
 *   fp32 (sqrt), labels int32; val <= 0 marks an empty slot.
 * Replaces BoxCoder.decode (vldyhead.py:78-108), clip_to_image, rpn/inference.py:696-708. */
int mq_box_decode(const float* val, const long* flat, const void* reg, int reg_f32, const float* anchors, const int* label_ids, long lab_bs,
                  const float* im_wh, float* boxes, float* scores, int* labels, int B, int K, int HW, int L,
                  long out_stride, long out_off, void* stream);

/* ROIAlign (legacy and aligned) -- the box pooler of the vision-query extraction path.
 *   feat: element (n, c, y, x) at feat + n*sn + c*sc + y*sh + x*sw (element strides; fp16, or fp32 when feat_f32 != 0),
 *   rois [R,5] fp32 (batch index, x1, y1, x2, y2), out [R,C,PH,PW] fp32 -- or [R,C] = mean over the PH*PW bins when
 *   reduce_mean != 0 (what extract_query keeps).  sampling_ratio <= 0: adaptive ceil(roi / bins) samples per bin.
 *   aligned != 0: torchvision's aligned=True (box shifted by -0.5 after scaling, no 1x1 minimum size).
 * Replaces _C.roi_align_forward (maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:16-123,262-306; layers/roi_align.py:13-57)
 *   and torchvision.ops.roi_align as called by ROIAlignV2 (layers/roi_align.py:71-81) inside Pooler / CustomPooler
 *   (modeling/poolers.py:45-168), used by GeneralizedVLRCNN_New.extract_query (generalized_vl_rcnn_new.py:232-288). */
int mq_roi_align_fwd(const void* feat, int feat_f32, const float* rois, float* out, int R, int C, int H, int W,
                     long sn, long sc, long sh, long sw, int PH, int PW, float spatial_scale, int sampling_ratio,
                     int aligned, int reduce_mean, void* stream);

/* Multi-scale deformable attention forward (Deformable-DETR / GroundingDINO):
 *   out[b,q,m*D+c] = sum_{l,p} attn[b,q,m,l,p] * bilinear(value[b, start_l.., m, c], loc[b,q,m,l,p,1]*H_l - 0.5, loc[..,0]*W_l - 0.5)
 *   value [B,S,M,D] fp16 (fp32 when value_f32), shapes [L,2] int64 (H, W), level_start [L] int64 (device pointers),
 *   loc [B,Q,M,L,P,2] fp32 in [0,1], attn [B,Q,M,L,P] fp32, out [B,Q,M*D] fp16 (fp32 when out_f32); D % 4 == 0.
 * Replaces groundingdino_new._C.ms_deform_attn_forward (csrc_groundingdino/vision.cpp:54, MsDeformAttn/ms_deform_attn_cuda.cu:21-81,
 *   ms_deform_im2col_cuda.cuh:33-84,237-299) and its torch fallback (ms_deform_attn.py:93-133). */
int mq_msdeform_attn_fwd(const void* value, int value_f32, const long* shapes, const long* level_start, const float* loc,
                         const float* attn, void* out, int out_f32, int B, int S, int M, int D, int L, int Q, int P, void* stream);

/* The same operator fed by the query-projection GEMM directly (softmax over the L*P samples and the sampling locations are
 * computed in registers; L == P == 4):
 *   qproj [B,Q,M*L*P*3] fp16 = [sampling_offsets (m,l,p,xy) | attention logits (m,l,p)] of one fused projection,
 *   ref [B,Q,L,ref_dim] fp32 normalised reference points (ref_dim 2: loc = ref + off / (W_l, H_l)) or boxes (ref_dim 4:
 *   loc = ref.xy + off / P * ref.wh * 0.5), value element (b,s,m,c) at value + b*value_bs + s*value_ts + m*D + c;
 *   valid_hw [B,L,2] int32 or NULL: un-padded (rows, columns) of every level per image -- corners outside read as zero, which
 *   is the reference's `value.masked_fill(key_padding_mask, 0)` (ms_deform_attn.py:286-287) without the masked copy.
 * Replaces MultiScaleDeformableAttention.forward lines ms_deform_attn.py:292-347 (view / softmax / location arithmetic /
 *   fp32 casts / _C.ms_deform_attn_forward) for the encoder (transformer.py:786-793) and decoder (:912-919) call sites. */
int mq_msdeform_attn_q_fwd(const void* value, int value_f32, long value_bs, long value_ts, const long* shapes,
                           const long* level_start, const void* qproj, const float* ref, int ref_dim, const int* valid_hw,
                           void* out, int out_f32, int B, int S, int M, int D, int L, int Q, int P, void* stream);

/* Class-aware NMS on score-sorted boxes, mask + sweep entirely on the device.
 *   boxes [B,N,4] fp32 (sorted by score desc per image), labels [B,N] int32, nvalid [B] int32 -> keep [B,N] uint8.
 * Replaces _C.ml_nms: maskrcnn_benchmark/csrc/ml_nms.h:10-27, csrc/cuda/ml_nms.cu:15-149 (vision.cpp:23). */
long mq_ml_nms_workspace_bytes(int B, int N);
int mq_ml_nms(const float* boxes, const int* labels, const int* nvalid, void* workspace, unsigned char* keep,
              int B, int N, float thr, void* stream);

/* The same NMS that stops sweeping an image once max_keep of its boxes are kept: with score-sorted input the first max_keep survivors
 * are the max_keep highest-scoring ones, all the caller uses (rpn/inference.py:757-766); keep[] is 0 behind the stopping chunk.
 * N <= 6656 (-1 beyond: use mq_ml_nms).  Default since round 3 (KERNELS["NMS_EARLY_STOP"] = 1). */
int mq_ml_nms_topk(const float* boxes, const int* labels, const int* nvalid, void* workspace, unsigned char* keep,
                   int B, int N, float thr, int max_keep, void* stream);

/* Swin PatchEmbed (4x4 stride-4 convolution as a 48 -> C projection, maskrcnn_benchmark/modeling/backbone/swint.py:447-471) +
 * patch_embed.norm + the first block's norm1 (swint.py:186-242) in one pass over the pixels:
 *   img: [B, Hi, Wi, 3] 16-bit channels-last pixels (img_f32 == 0) or the caller's [B, 3, Hi, Wi] fp32 tensor (img_f32 != 0: rounded to the
 *   operand type in the kernel -- no cast / layout pass before it); Hi, Wi multiples of 4.  w [C, 64] 16-bit = the conv weight in the k order
 *   of that layout (ops.patch_embed_pack; k 48..63 zero), bias / g0 / b0 (patch_embed.norm) / g1 / b1 (norm1 of layers.0.blocks.0) fp32 [C];
 *   x32 [B, Hi/4 * Wi/4, C] fp32 = LN_0(proj) (the residual stream), h1 (same shape, 16-bit) = LN_1(x32).  C in {96, 192} (-1 otherwise). */
int mq_patch_embed_fwd(const void* img, int img_f32, const void* w, const float* bias, const float* g0, const float* b0, const float* g1,
                       const float* b1, float* x32, void* h1, int B, int Hi, int Wi, int C, float eps, void* stream);

/* Self-attention over the text tokens with Q, K and V ROW-MAJOR (slices of one fused qkv projection): the second generation of
 * mq_attn_resident_fwd for the BERT layers (HF BertSelfAttention; the clamped copy rpn/modeling_bert.py:71-150 with clamp > 0).
 *   q [B,Nq,H,D], k / v [B,Nk,H,D] 16-bit views, unit last stride, strides in elements; o [B,Nq,H*D]; key_bias fp32 (b, h, j) at
 *   key_bias + b*bias_bs + h*bias_hs + j or NULL (<= -1e29: masked key); kv_len [B] int32 or NULL; max_kv: HOST bound on kv_len
 *   (0 = Nk): sizes the LDS tiles and picks the register variant (<= 160 keys: 10 key blocks per lane), must cover every kv_len[b].
 *   Nk <= 256, D in {32, 64}.  Returns -1 / -3 like mq_attn_resident_fwd. */
int mq_attn_text_fwd(const void* q, const void* k, const void* v, void* o, const float* key_bias, const int* kv_len,
                     int B, int H, int Nq, int Nk, int D, long q_bs, long q_rs, long q_hs, long k_bs, long k_rs, long k_hs,
                     long v_bs, long v_rs, long v_hs, long o_bs, long o_rs, long bias_bs, long bias_hs, float scale,
                     float clamp, int max_kv, void* stream);

/* ---- ATSS post-processing between the alignment kernel and the NMS in three launches (round 4; csrc/post2.hip).  fp32 / integer data only.
 * Replaces, inside ATSSPostProcessor (maskrcnn_benchmark/modeling/rpn/inference.py): forward_for_single_feature_map :677-712 (candidate
 * threshold mask, per-level topk(pre_nms_top_n), BoxCoder.decode vldyhead.py:78-108, clip_to_image, sqrt score, label ids) and
 * select_over_all_levels :757-766 (kthvalue + `>=` keeps ties with the K-th score) -- ~145 torch launches in one dependent chain.
 *
 * mq_post_select_fwd: per (image, level) the k[l] best candidates of the score map (value > 0 = candidate; exact radix select over the
 *   fp32 bit patterns, ties at the cut by the smaller flat index), decoded into the image's candidate list.  Two launches: the levels are
 *   cut into slices of 32768 scores whose k best go to a scratch list each, then one workgroup per (image, level) selects among them.
 *   ranked / reg / anchors: HOST arrays of NL device pointers -- level l: scores [B, hw[l], L] fp32, box deltas [B, hw[l], 4] fp32,
 *   anchors [hw[l], 4] fp32; hw / k: HOST int arrays (locations; candidates kept, 1 <= k[l] <= min(2048, hw[l] * L), hw[l] * L < 2^22);
 *   label_ids [L] int32 (lab_bs 0) or [B, L] (lab_bs L); im_wh [B, 2] fp32 (w, h); workspace: mq_post_select_workspace_bytes(...) bytes.
 *   Outputs (tot = sum k): boxes [B, tot, 4], scores [B, tot] (-1 = empty slot), labels [B, tot], ids [B, tot] int32 (candidate id = level
 *   base + flat index); level l owns slots [sum k[:l], sum k[:l + 1]), sorted by (score descending, flat index ascending), empty slots
 *   last.  -1: NL > 8 or a size out of range (mq_post_select_workspace_bytes returns -1 for the same inputs). */
long mq_post_select_workspace_bytes(const int* hw, const int* k, int NL, int B, int L);
int mq_post_select_fwd(const float* const* ranked, const float* const* reg, const float* const* anchors, const int* hw, const int* k,
                       int NL, int B, int L, const int* label_ids, long lab_bs, const float* im_wh, void* workspace, float* boxes,
                       float* scores, int* labels, int* ids, void* stream);
/* mq_post_sort_fwd: the NL per-level lists of every image (list l = slots [off[l], off[l + 1]), HOST ints, off[NL] = tot; each sorted by
 *   score descending with its empty slots last, as mq_post_select_fwd writes them) merged into ONE list ordered by (score descending,
 *   level ascending, position ascending), empty slots last -- the order ml_nms sweeps in (ml_nms.cu:100-104 sorts by score) -- + nvalid [B]
 *   = live rows.  tot <= 16384, NL <= 8 (-1 beyond). */
int mq_post_sort_fwd(const float* boxes, const float* scores, const int* labels, const int* off, int NL, float* boxes_o, float* scores_o,
                     int* labels_o, int* nvalid, int B, int tot, void* stream);
/* mq_post_finalize_fwd: rows score-sorted, keep [B, tot] uint8 from mq_ml_nms_topk(max_keep = K2) -> out [B, K2, 6] fp32 rows
 *   (x1, y1, x2, y2, score, label): the first K kept rows + the kept rows tied with the K-th score (inference.py:757-766), unused rows
 *   (0, 0, 0, 0, -1, 0); counts [B] int32 = live rows | 1 << 16 when all K2 - K tie slots hold ties (more may exist).  1 <= K <= K2 <= tot. */
int mq_post_finalize_fwd(const float* boxes, const float* scores, const int* labels, const unsigned char* keep, float* out,
                         int* counts, int B, int tot, int K, int K2, void* stream);

/* MFMA B-FRAGMENT ORDER of a weight matrix W [N, K] (N % 16 == 0, K % 32 == 0; ABI 28): the same elements as the row-major nn.Linear weight,
 * re-ordered ONCE when the checkpoint is loaded to [N / 16][K / 32][64][8] --
 *     packed[((n / 16) * (K / 32) + k / 32) * 512 + ((k % 32) / 8 * 16 + n % 16) * 8 + k % 8] = W[n][k]
 * (torch: W.view(N/16, 16, K/32, 4, 8).permute(0, 2, 3, 1, 4).contiguous(); mq_det_amd.ops.pack_b_fragments).  Lane l of a wave then finds its 8
 * operands of (16-column tile, k-step) at 8 l: one load instruction reads 1 KiB (fp32 twins: 2 KiB) of CONSECUTIVE bytes.  The two fused text
 * kernels below stream their weights L2 -> registers with no LDS stage; from the row-major matrix the same fragments are 16 rows x 64 B per
 * instruction, half a cache line per row, and stream at 37 GB/s per workgroup against 136 GB/s in this order (profiles/r05_l2_weight_stream.jsonl;
 * tools/probes/l2_weight_stream.hip).
 *
 * The attention half of a GatedCrossAttentionBlock in ONE launch (round 5; north-star "fused GCP + BERT attention"):
 *   x_out = x + tanh(w2 . gelu(Wg1 LN_g(sup))) * sup,  sup = Wout sparse_attn(Wq LN_a(x), kv, idx),  y = LN_f(x_out)
 * -- LayerNorm, to_q, the sparse gather-attention of mq_gcp_sparse_attn_fwd, to_out, the gate MLP with its LayerNorm, the gated residual of
 * mq_gcp_gate_residual_fwd and the LayerNorm in front of the feed-forward half; a workgroup owns 16 or 32 text rows, weights streamed from L2.
 * x / x_out [M, 768] fp32 residual stream (M = B*T text rows; x_out may alias x), y [M, 768] 16-bit or NULL, gate_out [M] fp32 or NULL
 * (VISION_QUERY.RETURN_ATTN_GATE_VALUE), kv [B, V, 1024] 16-bit = to_kv(norm_kv(vision)), idx [M, S] int32 (S <= 8, -1 = padding: a row
 * without a vision query gets sup == 0 exactly, quirk 5), wq [512, 768], wout [768, 512], wg1 [384, 768] IN B-FRAGMENT ORDER, w2 [384], the three LayerNorms'
 * gamma / beta [768] 16-bit; rows_per_block 16, 32 or 0 (chosen from M).  Returns -1 for other widths or S > 8.
 * Replaces GatedCrossAttentionBlock.forward up to the feed-forward half: maskrcnn_benchmark/modeling/language_backbone/modeling_bert_new.py
 * :298-368 (MaskedCrossAttention :162-248 with the sparse gather, attn_gate :340-359, the gated residual :368). */
int mq_gcp_attn_fwd(const float* x, float* x_out, void* y, float* gate_out, const void* kv, const int* idx, const void* wq, const void* wout,
                    const void* wg1, const void* w2, const void* ln_a_g, const void* ln_a_b, const void* ln_g_g, const void* ln_g_b,
                    const void* ln_f_g, const void* ln_f_b, long M, int T, int V, int S, int C, int heads, int dim_head, int G, float eps,
                    int rows_per_block, void* stream);

/* The attention half of a BERT layer in ONE launch (round 5; north-star "fused GCP + BERT attention"): the q | k | v projection of every
 * (batch item, head) AND its attention -- the qkv tensor is never written.  x [B, T, C] 16-bit hidden states (element (b, t, c) at
 * x + b*x_bs + t*x_rs + c; strides % 8 == 0), w [3C, C] the layer's fused projection weight (rows q | k | v) IN B-FRAGMENT ORDER (above), bias [3C],
 * o [B, T, C] context with the heads concatenated (o_rs % 4 == 0); key_bias fp32 (b, j) at key_bias + b*bias_bs + j or NULL (<= -1e29
 * marks a masked key); kv_len [B] int32 or NULL: keys at and beyond kv_len[b] are skipped in whole 16-key blocks.  C = 64 H, C % 128 == 0,
 * T <= 256 (one workgroup holds a whole (b, h): after the live-row compaction of the text T = 16 ceil(caption / 16)); clamp > 0 = the +-clamp
 * of the VLDyHead BERT copies, applied to the scaled logits before the mask as in the reference.  Returns -1 for other shapes, -3 for
 * misaligned strides.  Rounding points = the unfused path's (q, k, v rounded to 16 bits, P rounded for the second contraction).
 * Replaces BertSelfAttention.forward as a whole: query / key / value Linear + transpose_for_scores + matmul + mask + softmax + matmul
 * (HF modeling_bert.py BertSelfAttention; maskrcnn_benchmark/modeling/rpn/modeling_bert.py:71-170 with the clamps). */
int mq_bert_attn_qkv_fwd(const void* x, const void* w, const void* bias, void* o, const float* key_bias, const int* kv_len,
                         int B, int T, int C, int H, long x_bs, long x_rs, long o_bs, long o_rs, long bias_bs, float scale, float clamp,
                         void* stream);

/* ---- bf16 operands (BASELINE.json configs[3]: "MQ-GLIP-L ... bf16 MFMA").
 * Every entry point that reads or writes 16-bit operands exists twice: `name` as declared above (fp16, v_mfma_f32_16x16x32_f16) and
 * `name_bf16` -- the SAME kernel source compiled with bf16 operands (v_mfma_f32_16x16x32_bf16; fp32 accumulation, fp32 side inputs and
 * fp32 residual streams unchanged), same arguments, same return codes, "fp16" in the comments above read as "bf16".  The entry points
 * that only see fp32 / integer data (mq_abi_version, the *_workspace_bytes / mq_dcnv2_stats_blocks size queries, mq_ml_nms, mq_post_*) have no twin. */
#ifdef __cplusplus
#define MQ_BF16_TWIN(name) extern decltype(name) name##_bf16;
#else
#define MQ_BF16_TWIN(name) extern __typeof__(name) name##_bf16;
#endif
MQ_BF16_TWIN(mq_attn_fwd)
MQ_BF16_TWIN(mq_attn_resident_fwd)
MQ_BF16_TWIN(mq_attn_text_fwd)
MQ_BF16_TWIN(mq_bert_attn_qkv_fwd)
MQ_BF16_TWIN(mq_patch_embed_fwd)
MQ_BF16_TWIN(mq_attn_chunked_fwd)
MQ_BF16_TWIN(mq_window_attn_fwd)
MQ_BF16_TWIN(mq_window_attn_qkv_fwd)
MQ_BF16_TWIN(mq_gcp_sparse_attn_fwd)
MQ_BF16_TWIN(mq_gcp_gate_residual_fwd)
MQ_BF16_TWIN(mq_gcp_attn_fwd)
MQ_BF16_TWIN(mq_vlfuse_i2t_fwd)
MQ_BF16_TWIN(mq_vlfuse_t2i_fwd)
MQ_BF16_TWIN(mq_layernorm_fwd)
MQ_BF16_TWIN(mq_layernorm2_fwd)
MQ_BF16_TWIN(mq_layernorm_clamp_fwd)
MQ_BF16_TWIN(mq_clamp_gelu_clamp)
MQ_BF16_TWIN(mq_patch_merge_ln_fwd)
MQ_BF16_TWIN(mq_swin_mlp2_fwd)
MQ_BF16_TWIN(mq_conv3x3_fwd)
MQ_BF16_TWIN(mq_conv3x3_nchw32_fwd)
MQ_BF16_TWIN(mq_conv3x3_nchw32_v2_fwd)
MQ_BF16_TWIN(mq_conv3x3_nchw32_group_fwd)
MQ_BF16_TWIN(mq_dcnv2_fwd)
MQ_BF16_TWIN(mq_dcnv2_group_fwd)
MQ_BF16_TWIN(mq_dyconv_stats)
MQ_BF16_TWIN(mq_dyconv_coef)
MQ_BF16_TWIN(mq_dyconv_coef_group)
MQ_BF16_TWIN(mq_dyconv_fuse)
MQ_BF16_TWIN(mq_dyrelu_coef)
MQ_BF16_TWIN(mq_dyconv_epilogue_group)
MQ_BF16_TWIN(mq_dyrelu_apply)
MQ_BF16_TWIN(mq_add_upsample_nearest)
MQ_BF16_TWIN(mq_pool2x2_tokens_fwd)
MQ_BF16_TWIN(mq_dyrelu_ln_fwd)
MQ_BF16_TWIN(mq_align_scores_fwd)
MQ_BF16_TWIN(mq_align_fused_fwd)
MQ_BF16_TWIN(mq_box_decode)
MQ_BF16_TWIN(mq_roi_align_fwd)
MQ_BF16_TWIN(mq_msdeform_attn_fwd)
MQ_BF16_TWIN(mq_msdeform_attn_q_fwd)

/* ---- fp32 operands: the PRECISE mode (MODEL.COMPUTE_DTYPE = "float32"; BASELINE.json north_star: "outputs ... match the reference ...
 * within 1e-3").  `name_f32` is the SAME kernel source compiled a third time with every 16-bit operand a float (csrc/common.h under
 * -DMQ_F32: one 16x16x32 MFMA = THREE v_mfma_f32_16x16x32_f16 on the operands split as x = hi + lo / 2^11 (hi = fp16(x), lo = fp16((x - hi) 2^11):
 * ~22 operand bits, fp32 accumulation; round 5's eight v_mfma_f32_16x16x4_f32 are still there under -DMQ_F32_EXACT) on the same lane layout;
 * ds_read_b64_tr_b16 and the LDS-DMA copies replaced by their element-size-independent forms), same arguments and return codes with "fp16" read as "fp32" and every 16-bit LDS tile twice as
 * large -- a launch whose tiles exceed the 160 KB of a CU returns hipErrorInvalidValue (1); mq_det_amd/ops.py picks the shapes / variants
 * that fit.  3/16 of the fp16 MFMA rate and twice the bytes: this build is the configuration that meets the north-star's 1e-3 end to end (tests/, bench.py
 * `split_precise`).  No twin: mq_roi_align_fwd (its features may already be fp32 in the 16-bit builds: is_f32 flag). */
#ifdef __cplusplus
#define MQ_F32_TWIN(name) extern decltype(name) name##_f32;
#else
#define MQ_F32_TWIN(name) extern __typeof__(name) name##_f32;
#endif
/* Round 6 (ABI 29) -- what differs in the *_f32 entry points beside the element type:
 *   mq_swin_mlp2_fwd_f32   w1f / w2f are packed ALREADY SPLIT: every 512-element fragment block (2 KB) is [hi: 64 lanes x 8 fp16 | lo: 64 lanes x 8 fp16]
 *                          with hi = fp16(w), lo = fp16((w - hi) 2^11) (mq_det_amd.ops.split_planar_blocks; ops.swin_mlp2_pack does it in the precise mode);
 *   mq_dcnv2_*_f32, mq_vlfuse_*_f32, mq_conv3x3_nchw32_v2_fwd_f32   same arguments (fp32 tensors); inside, operands are split once when they are staged into
 *                          planar hi / lo fp16 LDS tiles (a k-step of the DCNv2 kernel is 32 channels there);
 *   mq_msdeform_attn_*_f32 new in ABI 29: `qproj` is fp32 (MQ-GroundingDINO in the precise mode).
 * Every other *_f32 entry splits its MFMA operands on the fly (csrc/common.h mfma16). */
MQ_F32_TWIN(mq_attn_fwd)
MQ_F32_TWIN(mq_attn_resident_fwd)
MQ_F32_TWIN(mq_attn_text_fwd)
MQ_F32_TWIN(mq_bert_attn_qkv_fwd)
MQ_F32_TWIN(mq_patch_embed_fwd)
MQ_F32_TWIN(mq_attn_chunked_fwd)
MQ_F32_TWIN(mq_window_attn_fwd)
MQ_F32_TWIN(mq_window_attn_qkv_fwd)
MQ_F32_TWIN(mq_gcp_sparse_attn_fwd)
MQ_F32_TWIN(mq_gcp_gate_residual_fwd)
MQ_F32_TWIN(mq_gcp_attn_fwd)
MQ_F32_TWIN(mq_vlfuse_i2t_fwd)
MQ_F32_TWIN(mq_vlfuse_t2i_fwd)
MQ_F32_TWIN(mq_layernorm_fwd)
MQ_F32_TWIN(mq_layernorm2_fwd)
MQ_F32_TWIN(mq_layernorm_clamp_fwd)
MQ_F32_TWIN(mq_clamp_gelu_clamp)
MQ_F32_TWIN(mq_patch_merge_ln_fwd)
MQ_F32_TWIN(mq_swin_mlp2_fwd)
MQ_F32_TWIN(mq_conv3x3_fwd)
MQ_F32_TWIN(mq_conv3x3_nchw32_fwd)
MQ_F32_TWIN(mq_conv3x3_nchw32_v2_fwd)
MQ_F32_TWIN(mq_conv3x3_nchw32_group_fwd)
MQ_F32_TWIN(mq_dcnv2_fwd)
MQ_F32_TWIN(mq_dcnv2_group_fwd)
MQ_F32_TWIN(mq_dyconv_stats)
MQ_F32_TWIN(mq_dyconv_coef)
MQ_F32_TWIN(mq_dyconv_coef_group)
MQ_F32_TWIN(mq_dyconv_fuse)
MQ_F32_TWIN(mq_dyrelu_coef)
MQ_F32_TWIN(mq_dyconv_epilogue_group)
MQ_F32_TWIN(mq_dyrelu_apply)
MQ_F32_TWIN(mq_dyrelu_ln_fwd)
MQ_F32_TWIN(mq_add_upsample_nearest)
MQ_F32_TWIN(mq_pool2x2_tokens_fwd)
MQ_F32_TWIN(mq_align_scores_fwd)
MQ_F32_TWIN(mq_align_fused_fwd)
MQ_F32_TWIN(mq_box_decode)
MQ_F32_TWIN(mq_msdeform_attn_fwd)
MQ_F32_TWIN(mq_msdeform_attn_q_fwd)

#ifdef __cplusplus
}
#endif
#endif
