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

MQ_NAMESPACE_BEGIN

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

// Any S: online softmax over the slots, one at a time (SELECT_FPN_LEVEL = False stores 5 scales per query: S = 25).
__global__ __launch_bounds__(256) void gcp_sparse_attn_loop_kernel(const half_t* __restrict__ q, const half_t* __restrict__ kv,
                                                                   const int* __restrict__ idx, half_t* __restrict__ out,
                                                                   int B, int T, int V, int S, float scale) {
  const int lane = threadIdx.x & 63;
  const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= (long)B * T) return;
  const int b = tok / T;
  constexpr int HD = 512;
  const half8 qv = *(const half8*)(q + tok * HD + lane * 8);
  float qf[8], acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { qf[j] = (float)qv[j] * scale; acc[j] = 0.f; }
  float mx = MQ_NEG_BIG, den = 0.f;
  for (int s = 0; s < S; ++s) {
    const int id = idx[tok * S + s];
    if (id < 0) continue;                         // wave-uniform (one token per wave)
    const half_t* row = kv + ((long)b * V + id) * (2 * HD);
    const half8 kk = *(const half8*)(row + lane * 8), vv = *(const half8*)(row + HD + lane * 8);
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) d += qf[j] * (float)kk[j];
    d += __shfl_xor(d, 1);
    d += __shfl_xor(d, 2);
    d += __shfl_xor(d, 4);
    const float nm = fmaxf(mx, d), c = __expf(mx - nm), e = __expf(d - nm);
    den = den * c + e;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = acc[j] * c + e * (float)vv[j];
    mx = nm;
  }
  const float inv = den > 0.f ? 1.f / den : 0.f;   // no valid slot: exactly zero (quirk 5)
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (half_t)(acc[j] * inv);
  *(half8*)(out + tok * HD + lane * 8) = o;
}

extern "C" int MQ_SYM(mq_gcp_sparse_attn_fwd)(const void* q, const void* kv, const int* idx, void* out, int B, int T, int V,
                                      int S, int heads, int dim_head, void* stream) {
  if (B <= 0 || T <= 0) return 0;
  if (heads * dim_head != 512 || dim_head != 64 || S < 0) return -1;
  float scale = 1.0f / sqrtf((float)dim_head);
  dim3 grid((unsigned)(((long)B * T + 3) / 4));
  if (S <= 8)
    hipLaunchKernelGGL((gcp_sparse_attn_kernel<8>), grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)q,
                       (const half_t*)kv, idx, (half_t*)out, B, T, V, S, scale);
  else if (S <= 16)
    hipLaunchKernelGGL((gcp_sparse_attn_kernel<16>), grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)q,
                       (const half_t*)kv, idx, (half_t*)out, B, T, V, S, scale);
  else
    hipLaunchKernelGGL(gcp_sparse_attn_loop_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)q,
                       (const half_t*)kv, idx, (half_t*)out, B, T, V, S, scale);
  MQ_CHECK_LAUNCH();
  return 0;
}

template <bool XF32>
__global__ __launch_bounds__(256) void gcp_gate_residual_kernel(const half_t* __restrict__ sup, const half_t* __restrict__ h,
                                                                const half_t* __restrict__ w2, const void* __restrict__ x,
                                                                void* __restrict__ out, float* __restrict__ gate_out,
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
    if constexpr (XF32) {
      const float* xr = (const float*)x + row * C + i;
      float* o = (float*)out + row * C + i;
      o[0] = (float)s[0] * gate + xr[0];
      o[1] = (float)s[1] * gate + xr[1];
    } else {
      half2_ r = *(const half2_*)((const half_t*)x + row * C + i);
      half2_ o;
      o[0] = (half_t)((float)s[0] * gate + (float)r[0]);
      o[1] = (half_t)((float)s[1] * gate + (float)r[1]);
      *(half2_*)((half_t*)out + row * C + i) = o;
    }
  }
}

extern "C" int MQ_SYM(mq_gcp_gate_residual_fwd)(const void* sup, const void* h, const void* w2, const void* x, int x_f32, void* out,
                                        float* gate_out, long M, int C, int G, void* stream) {
  if (M <= 0) return 0;
  if (C % 2) return -1;
  if (x_f32)
    hipLaunchKernelGGL(gcp_gate_residual_kernel<true>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)sup, (const half_t*)h, (const half_t*)w2, x, out, gate_out, M, C, G);
  else
    hipLaunchKernelGGL(gcp_gate_residual_kernel<false>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)sup, (const half_t*)h, (const half_t*)w2, x, out, gate_out, M, C, G);
  MQ_CHECK_LAUNCH();
  return 0;
}

MQ_NAMESPACE_END
