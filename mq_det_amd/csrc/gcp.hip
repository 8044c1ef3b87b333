// Gated Class-scalable Perceiver (GCP) kernels for gfx950.
//
// mq_gcp_sparse_attn_fwd -- the "sparse" MaskedCrossAttention of the reference
//   (language_backbone/modeling_bert_new.py:162-184 gather, :204-240 attention): every text token
//   attends only to the <= S vision-query tokens of its own category.  The reference gathers S copies
//   of the vision rows per text token and re-projects them (B*T*S K/V projections); here K/V are
//   projected ONCE per unique vision token (plain GEMM outside) and the kernel gathers rows by index.
//     q   : [B, T, heads*64] fp16 = to_q(norm(x))            (unscaled; scale applied here)
//     kv  : [B, V, 2*heads*64] fp16 = to_kv(norm_kv(vision)) (k | v)
//     idx : [B, T, S] int32, indices into V, -1 = padding (reference: index V -> appended zero row)
//     out : [B, T, heads*64] fp16
//   Semantics kept bit-for-intent: additive -1e4 on padded slots then softmax, then attn *= mask, so a
//   token with no vision query gets EXACTLY zero (quirk 5); for tokens with >= 1 query the padded slots
//   contribute exp(-1e4 + ...) == 0 in fp32, i.e. softmax over the valid slots only.
//   One wave per (b, t): lane owns 8 consecutive channels (64 lanes x 8 = 512 = 8 heads x 64), the 8
//   lanes of a head reduce q.k with wave shuffles.  HBM-bound gather, no MFMA (K = 64, <= 8 keys).
//
// mq_gcp_gate_residual_fwd -- the conditional gate fused into the residual
//   (modeling_bert_new.py:359,368):  x_out = sup * tanh( w2 . gelu(h) ) + x
//     sup : [M, C] fp16 cross-attention output (after to_out), h : [M, G] fp16 = linear1(norm(sup)),
//     w2  : [G] fp16 (attn_gate.linear2.weight), x : [M, C] fp16 residual stream.
//   One wave per row: GELU(erf) + dot + tanh + axpy in one pass.
#include "common.h"

template <int SMAX>
__global__ __launch_bounds__(256) void gcp_sparse_attn_kernel(const half_t* __restrict__ q, const half_t* __restrict__ kv,
                                                              const int* __restrict__ idx, half_t* __restrict__ out,
                                                              int B, int T, int V, int S, float scale) {
  const int lane = threadIdx.x & 63;
  const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= (long)B * T) return;
  const int b = tok / T;
  constexpr int HD = 512;                       // heads * dim_head
  half8 qv = *(const half8*)(q + tok * HD + lane * 8);
  float qf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) qf[j] = (float)qv[j] * scale;
  float sim[SMAX];
  half8 vv[SMAX];
  bool any = false;
  float mx = MQ_NEG_BIG;
#pragma unroll
  for (int s = 0; s < SMAX; ++s) {
    int id = s < S ? idx[tok * S + s] : -1;
    sim[s] = MQ_NEG_BIG;
    vv[s] = zero8();
    if (id >= 0) {
      const half_t* row = kv + ((long)b * V + id) * (2 * HD);
      half8 kk = *(const half8*)(row + lane * 8);
      vv[s] = *(const half8*)(row + HD + lane * 8);
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) d += qf[j] * (float)kk[j];
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      d += __shfl_xor(d, 4);                   // 8 lanes of this head
      sim[s] = d;
      mx = fmaxf(mx, d);
      any = true;
    }
  }
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (any) {
    float den = 0.f;
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      float e = sim[s] > 0.5f * MQ_NEG_BIG ? __expf(sim[s] - mx) : 0.f;
      sim[s] = e;
      den += e;
    }
    float inv = 1.f / den;
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      float w = sim[s] * inv;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += w * (float)vv[s][j];
    }
  }
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
  *(half8*)(out + tok * HD + lane * 8) = o;
}

extern "C" int mq_gcp_sparse_attn_fwd(const void* q, const void* kv, const int* idx, void* out, int B, int T, int V,
                                      int S, int heads, int dim_head, void* stream) {
  if (B <= 0 || T <= 0) return 0;
  if (heads * dim_head != 512 || dim_head != 64 || S > 16 || S < 0) return -1;
  float scale = 1.0f / sqrtf((float)dim_head);
  dim3 grid((unsigned)(((long)B * T + 3) / 4));
  if (S <= 8)
    hipLaunchKernelGGL((gcp_sparse_attn_kernel<8>), grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)q,
                       (const half_t*)kv, idx, (half_t*)out, B, T, V, S, scale);
  else
    hipLaunchKernelGGL((gcp_sparse_attn_kernel<16>), grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)q,
                       (const half_t*)kv, idx, (half_t*)out, B, T, V, S, scale);
  MQ_CHECK_LAUNCH();
  return 0;
}

__global__ __launch_bounds__(256) void gcp_gate_residual_kernel(const half_t* __restrict__ sup, const half_t* __restrict__ h,
                                                                const half_t* __restrict__ w2, const half_t* __restrict__ x,
                                                                half_t* __restrict__ out, float* __restrict__ gate_out,
                                                                long M, int C, int G) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float d = 0.f;
  for (int i = lane; i < G; i += 64) {
    float v = (float)h[row * G + i];
    float g = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));      // exact (erf) GELU, nn.GELU()
    d += g * (float)w2[i];
  }
  d = wave_sum(d);
  float gate = tanhf(d);
  if (gate_out && lane == 0) gate_out[row] = gate;
  for (int i = lane * 2; i < C; i += 128) {
    half2_ s = *(const half2_*)(sup + row * C + i);
    half2_ r = *(const half2_*)(x + row * C + i);
    half2_ o;
    o[0] = (half_t)((float)s[0] * gate + (float)r[0]);
    o[1] = (half_t)((float)s[1] * gate + (float)r[1]);
    *(half2_*)(out + row * C + i) = o;
  }
}

extern "C" int mq_gcp_gate_residual_fwd(const void* sup, const void* h, const void* w2, const void* x, void* out,
                                        float* gate_out, long M, int C, int G, void* stream) {
  if (M <= 0) return 0;
  if (C % 2) return -1;
  hipLaunchKernelGGL(gcp_gate_residual_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)sup, (const half_t*)h, (const half_t*)w2, (const half_t*)x, (half_t*)out, gate_out,
                     M, C, G);
  MQ_CHECK_LAUNCH();
  return 0;
}

// mq_headsum_residual_fwd:  out[m, c] = res[m, c] + bias[c] + sum_h x[m, h, c]      (fp16, fp32 accumulate)
// VLFuse image side: with out_v_proj (and the layer scale) folded into the text-side values, the per-head attention
// outputs only need to be summed over the 8 heads and added to the residual LN(v) -- this replaces the
// [B*22400, 2048] x [2048, 256] out_v_proj GEMM (1.6 ms / step) by one HBM-bound pass (fuse_helper.py:300,424).
__global__ __launch_bounds__(256) void headsum_residual_kernel(const half_t* __restrict__ x, const half_t* __restrict__ res,
                                                               const half_t* __restrict__ bias, half_t* __restrict__ out,
                                                               long M, int H, int C) {
  const int cpt = C / 8;
  const long total = M * cpt;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / cpt;
    const int c0 = (int)(i % cpt) * 8;
    half8 r = *(const half8*)(res + m * C + c0), bb = *(const half8*)(bias + c0);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (float)r[j] + (float)bb[j];
    for (int h = 0; h < H; ++h) {
      half8 v = *(const half8*)(x + (m * H + h) * C + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
    *(half8*)(out + m * C + c0) = o;
  }
}

extern "C" int mq_headsum_residual_fwd(const void* x, const void* res, const void* bias, void* out, long M, int H, int C,
                                       void* stream) {
  if (M <= 0) return 0;
  if (C % 8) return -1;
  long blocks = (M * (C / 8) + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(headsum_residual_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const half_t*)x,
                     (const half_t*)res, (const half_t*)bias, (half_t*)out, M, H, C);
  MQ_CHECK_LAUNCH();
  return 0;
}
