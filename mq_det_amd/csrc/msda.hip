// mq_msdeform_attn_fwd: multi-scale deformable attention forward (Deformable-DETR / GroundingDINO) for gfx950.
//
//   out[b, q, m*D + c] = sum_{l < L, p < P} attn[b,q,m,l,p] * bilinear( value[b, start_l : start_l + H_l W_l, m, c],
//                                                                         loc[b,q,m,l,p,1] * H_l - 0.5, loc[..,0] * W_l - 0.5 )
//   (zero padding, align_corners = False; a sample contributes only if -1 < h < H_l and -1 < w < W_l)
//
// Reference: groundingdino_new/models/GroundingDINO/csrc_groundingdino/MsDeformAttn/ms_deform_im2col_cuda.cuh:33-84
// (ms_deform_attn_im2col_bilinear), :237-299 (ms_deformable_im2col_gpu_kernel), host wrapper ms_deform_attn_cuda.cu:21-81,
// Python fallback ms_deform_attn.py:93-133.  The reference runs it in fp32 (ms_deform_attn.py:330-336) with ONE THREAD PER
// OUTPUT ELEMENT: the 32 channels of a head sit in 32 different threads that each re-read the same sampling location and
// weight and issue 4-byte loads.  MI355X shape: the op is a pure gather (11 GF of MACs against 22 323 x 8 x 16 random 64-byte
// reads per image): one wave owns one (b, q); lane = (head, 4-channel group), so the D/4 lanes of a head read
// one contiguous D-channel row per corner (64 B in fp16 for D = 32 -- one request), sampling state is computed once per
// lane group, value may be fp16 (the value_proj GEMM's output dtype) with fp32 weights / accumulation.
//   value  [B, S, M, D]  fp16 or fp32      shapes [L, 2] int64 (H, W)      level_start [L] int64
//   loc    [B, Q, M, L, P, 2] fp32 (x, y in [0, 1])      attn [B, Q, M, L, P] fp32
//   out    [B, Q, M * D]  fp16 or fp32
#include "common.h"
#include <cstdlib>

MQ_NAMESPACE_BEGIN

template <typename TV, typename TO>
__global__ __launch_bounds__(256) void msda_kernel(const TV* __restrict__ value, const long* __restrict__ shapes,
                                                   const long* __restrict__ level_start, const float* __restrict__ loc,
                                                   const float* __restrict__ attn, TO* __restrict__ out, int B, int S, int M, int D,
                                                   int L, int Q, int P) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int bq = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));     // wave-uniform: row pointers in SGPRs
  if (bq >= B * Q) return;
  const int b = bq / Q;
  const int lph = D >> 2;                                   // lanes per head (4 channels per lane)
  const int nchunk = M * lph;                               // 4-channel groups of one output row
  for (int ch = lane; ch < nchunk; ch += 64) {
    const int m = ch / lph, c = (ch - m * lph) << 2;
    const float* lp = loc + (((long)bq * M + m) * L) * P * 2;
    const float* ap = attn + (((long)bq * M + m) * L) * P;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const TV* vb = value + ((long)b * S + level_start[l]) * M * D;
      const unsigned ws = (unsigned)(M * D), hs = (unsigned)W * ws, lane_off = (unsigned)(m * D + c);
      for (int pt = 0; pt < P; ++pt) {
        // branch-free: the four corner addresses are clamped into the level and loaded unconditionally (4 loads in flight
        // instead of 4 guarded, serialised ones), corners outside the map get weight 0
        const float loc_w = lp[(l * P + pt) * 2], loc_h = lp[(l * P + pt) * 2 + 1];
        float wgt = ap[l * P + pt];
        const float h = loc_h * (float)H - 0.5f, w = loc_w * (float)W - 0.5f;
        if (!(h > -1.f && w > -1.f && h < (float)H && w < (float)W)) wgt = 0.f;
        const float hf = floorf(h), wf = floorf(w);
        const int hl = (int)hf, wl = (int)wf, hh = hl + 1, wh = wl + 1;
        const float lh = h - hf, lw = w - wf, uh = 1.f - lh, uw = 1.f - lw;
        const bool y0 = hl >= 0, y1 = hh <= H - 1, x0 = wl >= 0, x1 = wh <= W - 1;
        const float cw[4] = {(y0 && x0) ? uh * uw * wgt : 0.f, (y0 && x1) ? uh * lw * wgt : 0.f,
                             (y1 && x0) ? lh * uw * wgt : 0.f, (y1 && x1) ? lh * lw * wgt : 0.f};
        const unsigned yc0 = (unsigned)min(max(hl, 0), H - 1) * hs, yc1 = (unsigned)min(max(hh, 0), H - 1) * hs;
        const unsigned xc0 = (unsigned)min(max(wl, 0), W - 1) * ws, xc1 = (unsigned)min(max(wh, 0), W - 1) * ws;
        const unsigned ca[4] = {yc0 + xc0 + lane_off, yc0 + xc1 + lane_off, yc1 + xc0 + lane_off, yc1 + xc1 + lane_off};
        if constexpr (sizeof(TV) == 2) {
          half4 cv[4];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) cv[q4] = *(const half4*)(vb + ca[q4]);
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += cw[q4] * (float)cv[q4][j];
        } else {
          float4_ cv[4];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) cv[q4] = *(const float4_*)(vb + ca[q4]);
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += cw[q4] * cv[q4][j];
        }
      }
    }
    TO* o = out + (long)bq * (M * D) + m * D + c;
    if constexpr (sizeof(TO) == 2) {
      half4 t;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = (half_t)acc[j];
      *(half4*)o = t;
    } else {
      *(float4_*)o = (float4_){acc[0], acc[1], acc[2], acc[3]};
    }
  }
}

extern "C" int MQ_SYM(mq_msdeform_attn_fwd)(const void* value, int value_f32, const long* shapes, const long* level_start, const float* loc,
                                    const float* attn, void* out, int out_f32, int B, int S, int M, int D, int L, int Q, int P,
                                    void* stream) {
  if (B <= 0 || Q <= 0) return 0;
  if (D % 4 || M <= 0 || L <= 0 || P <= 0) return -1;
  const dim3 grid((unsigned)(((long)B * Q + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
#define MQ_MSDA(TV, TO)                                                                                                 \
  hipLaunchKernelGGL((msda_kernel<TV, TO>), grid, dim3(256), 0, s, (const TV*)value, shapes, level_start, loc, attn, (TO*)out, B, S, \
                     M, D, L, Q, P)
  if (value_f32 && out_f32) MQ_MSDA(float, float);
  else if (value_f32) MQ_MSDA(float, half_t);
  else if (out_f32) MQ_MSDA(half_t, float);
  else MQ_MSDA(half_t, half_t);
#undef MQ_MSDA
  MQ_CHECK_LAUNCH();
  return 0;
}


// ------------------------------------------------------------------------------------------------ fused query side
// mq_msdeform_attn_q_fwd: the same gather, fed with what the query projection GEMM produced instead of materialised fp32
// sampling locations / attention weights (reference ms_deform_attn.py:292-329 builds them with a softmax, two divisions, a
// broadcast add and several fp32 casts: ~8 elementwise passes over [B, Q, M, L, P, 2] per layer).  Per (b, q, head):
//   logits -> softmax over the L*P samples, in registers;
//   loc = ref[l, :2] + off / (W_l, H_l)                      (2-d reference points, encoder)
//       = ref[l, :2] + off / P * ref[l, 2:] * 0.5            (4-d reference boxes, decoder)
//   qproj [B, Q, M*L*P*3] fp16: offsets (m, l, p, xy) then logits (m, l, p) -- the [sampling_offsets | attention_weights]
//   projection as ONE GEMM; ref [B, Q, L, RD] fp32; value element (b, s, m, c) at value + b*value_bs + s*value_ts + m*D + c
//   (a token stride > M*D lets the six decoder layers share one batched value projection).
//   valid_hw [B, L, 2] int32 or nullptr: (rows, columns) of level l that are NOT padding for image b.  The reference zeroes the
//   value rows of padding tokens (`value.masked_fill(key_padding_mask, 0)`, ms_deform_attn.py:286-287); the padding of a
//   batched image is the region right of / below its valid rectangle, so a corner outside the rectangle simply reads as zero
//   here -- no masked copy of the [B, S, 256] value tensor per layer.
template <typename TV, typename TO, int L, int P, int RD>
__global__ __launch_bounds__(256) void msda_q_kernel(const TV* __restrict__ value, long value_bs, long value_ts,
                                                     const long* __restrict__ shapes, const long* __restrict__ level_start,
                                                     const half_t* __restrict__ qproj, const float* __restrict__ ref,
                                                     const int* __restrict__ valid_hw, TO* __restrict__ out, int B, int S, int M,
                                                     int D, int Q, int xcd_order) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // XCD-aware order: workgroup i runs on XCD i % 8 and every XCD has its own 4 MB L2.  With the natural order each XCD sees
  // every 8th group of 4 queries, i.e. ALL value rows of ALL images stream through every L2 (11.4 MB of fp16 values per
  // 800 x 1344 image) and most corner reads miss it.  Here XCD x owns the x-th CONTIGUOUS eighth of the (image, query) range:
  // queries that run at the same time on one XCD are spatial neighbours and sample overlapping value rows.
  const long nblk = ((long)B * Q + 3) / 4;
  long blk = blockIdx.x;
  if (xcd_order) {
    const long per = (nblk + 7) / 8;
    blk = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per) return;
  }
  // wave-uniform by construction: as scalars, the row pointers below live in SGPRs and the gathers use 32-bit lane offsets
  const int bq = __builtin_amdgcn_readfirstlane((int)(blk * 4 + wave));
  if (blk >= nblk || bq >= B * Q) return;
  const int b = bq / Q;
  const int lph = D >> 2;
  const int nchunk = M * lph;
  constexpr int LP = L * P;
  const half_t* qrow = qproj + (long)bq * (M * LP * 3);
  const float* rrow = ref + (long)bq * (L * RD);
  for (int ch = lane; ch < nchunk; ch += 64) {
    const int m = ch / lph, c = (ch - m * lph) << 2;
    // softmax over this head's L*P logits: max and 1/sum here (two 16-byte loads, shared by the lanes of the head through L1),
    // the weights themselves are formed per level below -- no 16-entry register array, the kernel stays at 8 waves / SIMD
    const half_t* lg = qrow + M * LP * 2 + m * LP;
    float mx = -3.0e38f, inv;
    {
      float e[LP];
#pragma unroll
      for (int i = 0; i < LP; i += 8) {
        const half8 t = *(const half8*)(lg + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) { e[i + j] = (float)t[j]; mx = fmaxf(mx, e[i + j]); }
      }
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < LP; ++i) sum += __expf(e[i] - mx);
      inv = 1.f / sum;
    }
    const half_t* op = qrow + m * LP * 2;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1                                             // one level at a time: 16 corner loads in flight, <= 64 VGPRs (8 waves / SIMD)
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const int Hv = valid_hw ? valid_hw[(b * L + l) * 2] : H, Wv = valid_hw ? valid_hw[(b * L + l) * 2 + 1] : W;
      const TV* vb = value + (long)b * value_bs + level_start[l] * value_ts;      // uniform; lane part: m * D + c + (y W + x) ts
      const unsigned ws = (unsigned)value_ts, hs = (unsigned)W * ws, lane_off = (unsigned)(m * D + c);
      const float rx = rrow[l * RD], ry = rrow[l * RD + 1];
      float sx, sy;                                        // offset -> normalised location scale
      if constexpr (RD == 2) { sx = 1.f / (float)W; sy = 1.f / (float)H; }
      else { sx = rrow[l * RD + 2] * (0.5f / (float)P); sy = rrow[l * RD + 3] * (0.5f / (float)P); }
      const half8 o8 = *(const half8*)(op + l * P * 2);     // P == 4: the 4 (x, y) pairs of this level
      const half4 l4 = *(const half4*)(lg + l * P);         // ... and their 4 logits
      // branch-free gather: every corner address is clamped into the level and loaded unconditionally, corners outside the
      // map / the valid rectangle get weight 0.  (With one guarded load per corner the compiler emitted branch + s_waitcnt
      // vmcnt(0) after each of them -- 64 serialised L2 round trips per query, 14.3 ms per forward at B = 16.)
      float cw[P][4];
      unsigned ca[P][4];
#pragma unroll
      for (int pt = 0; pt < P; ++pt) {
        const float loc_w = rx + (float)o8[2 * pt] * sx, loc_h = ry + (float)o8[2 * pt + 1] * sy;
        float wgt = __expf((float)l4[pt] - mx) * inv;
        const float h = loc_h * (float)H - 0.5f, wq = loc_w * (float)W - 0.5f;
        if (!(h > -1.f && wq > -1.f && h < (float)H && wq < (float)W)) wgt = 0.f;
        const float hf = floorf(h), wf = floorf(wq);
        const int hl = (int)hf, wl = (int)wf, hh = hl + 1, wh = wl + 1;
        const float lh = h - hf, lw = wq - wf, uh = 1.f - lh, uw = 1.f - lw;
        const bool y0 = hl >= 0 && hl < Hv, y1 = hh >= 0 && hh < Hv, x0 = wl >= 0 && wl < Wv, x1 = wh >= 0 && wh < Wv;
        cw[pt][0] = (y0 && x0) ? uh * uw * wgt : 0.f;
        cw[pt][1] = (y0 && x1) ? uh * lw * wgt : 0.f;
        cw[pt][2] = (y1 && x0) ? lh * uw * wgt : 0.f;
        cw[pt][3] = (y1 && x1) ? lh * lw * wgt : 0.f;
        const unsigned yc0 = (unsigned)min(max(hl, 0), H - 1) * hs, yc1 = (unsigned)min(max(hh, 0), H - 1) * hs;
        const unsigned xc0 = (unsigned)min(max(wl, 0), W - 1) * ws, xc1 = (unsigned)min(max(wh, 0), W - 1) * ws;
        ca[pt][0] = yc0 + xc0 + lane_off; ca[pt][1] = yc0 + xc1 + lane_off;
        ca[pt][2] = yc1 + xc0 + lane_off; ca[pt][3] = yc1 + xc1 + lane_off;
      }
      if constexpr (sizeof(TV) == 2) {
        half4 cv[P][4];
#pragma unroll
        for (int pt = 0; pt < P; ++pt)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) cv[pt][q4] = *(const half4*)(vb + ca[pt][q4]);
#pragma unroll
        for (int pt = 0; pt < P; ++pt)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += cw[pt][q4] * (float)cv[pt][q4][j];
      } else {
        float4_ cv[P][4];
#pragma unroll
        for (int pt = 0; pt < P; ++pt)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) cv[pt][q4] = *(const float4_*)(vb + ca[pt][q4]);
#pragma unroll
        for (int pt = 0; pt < P; ++pt)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += cw[pt][q4] * cv[pt][q4][j];
      }
    }
    TO* o = out + (long)bq * (M * D) + m * D + c;
    if constexpr (sizeof(TO) == 2) {
      half4 t;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = (half_t)acc[j];
      *(half4*)o = t;
    } else {
      *(float4_*)o = (float4_){acc[0], acc[1], acc[2], acc[3]};
    }
  }
}

extern "C" int MQ_SYM(mq_msdeform_attn_q_fwd)(const void* value, int value_f32, long value_bs, long value_ts, const long* shapes,
                                      const long* level_start, const void* qproj, const float* ref, int ref_dim,
                                      const int* valid_hw, void* out, int out_f32, int B, int S, int M, int D, int L, int Q, int P,
                                      void* stream) {
  if (B <= 0 || Q <= 0) return 0;
  if (D % 4 || M <= 0 || L != 4 || P != 4 || (ref_dim != 2 && ref_dim != 4) || (value_ts % 4) || (value_bs % 4)) return -1;
  // A/B switch.  Measured (GPU call 10, B = 16): 6.22 ms with the XCD-contiguous order, 6.10 ms with the natural one -- after the
  // branch-free gather the kernel is bound by the rate of 64-byte requests (18 B/clk/CU), not by L2 misses: natural order.
  static const int xcd_order = [] { const char* e = getenv("MQ_MSDA_ORDER"); return (e && e[0] == '1') ? 1 : 0; }();
  const long nblk = ((long)B * Q + 3) / 4;
  const dim3 grid((unsigned)(xcd_order ? 8 * ((nblk + 7) / 8) : nblk));
  hipStream_t s = (hipStream_t)stream;
#define MQ_MSDAQ(TV, TO, RD)                                                                                             \
  hipLaunchKernelGGL((msda_q_kernel<TV, TO, 4, 4, RD>), grid, dim3(256), 0, s, (const TV*)value, value_bs, value_ts, shapes,  \
                     level_start, (const half_t*)qproj, ref, valid_hw, (TO*)out, B, S, M, D, Q, xcd_order)
#define MQ_MSDAQ_T(RD)                                       \
  if (value_f32 && out_f32) MQ_MSDAQ(float, float, RD);      \
  else if (value_f32) MQ_MSDAQ(float, half_t, RD);           \
  else if (out_f32) MQ_MSDAQ(half_t, float, RD);             \
  else MQ_MSDAQ(half_t, half_t, RD)
  if (ref_dim == 2) { MQ_MSDAQ_T(2); } else { MQ_MSDAQ_T(4); }
#undef MQ_MSDAQ_T
#undef MQ_MSDAQ
  MQ_CHECK_LAUNCH();
  return 0;
}

MQ_NAMESPACE_END
