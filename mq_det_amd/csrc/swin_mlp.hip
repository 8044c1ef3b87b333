// mq_swin_mlp_fwd: the MLP half of a Swin block as ONE kernel for gfx950:
//
//     x'   = x + delta                       (delta = attention projection output, swint.py:236; optional)
//     out  = x' + fc2( GELU( fc1( LayerNorm(x') ) ) )        (swint.py:240, Mlp :13-31, exact erf GELU)
//     y    = LayerNorm_next(out)             (optional: the next block's norm1 / the stage's output norm, fp16)
//
// Reference path: norm2 -> fc1 -> GELU -> fc2 -> residual = 4 full-tensor kernels around two GEMMs; the 4C-wide hidden
// activation (stage 1, B = 8: 413 MB in fp16) is written and re-read three times.  Round 1 of this repository ran the two
// GEMMs in hipBLASLt with a torch GELU in between: at K = 96 / 192 those GEMMs are HBM-bound (115 us for 516 MB), the
// GELU pass alone was 1.0 ms / forward.  Here the hidden activation never leaves the register file:
//
//   * a workgroup owns 128 tokens (8 waves x 16); the residual stream x is fp32 in HBM (read once, written once);
//   * LayerNorm is the prologue: each wave normalises its 16 rows, parks them as fp16 in LDS and pulls them back as
//     MFMA B-fragments that stay in registers for the whole kernel;
//   * both GEMMs are computed TRANSPOSED:  H^T = W1 . LN(x)^T  and  OUT^T = W2 . H^T.  With v_mfma_f32_16x16x32_f16 the
//     accumulator of the first product (lane holds 4 consecutive hidden units of one token) is, after bias + GELU + cvt,
//     exactly a B-fragment of the second one -- provided the k-slots of W2 are permuted accordingly (done once on the
//     host: slot 8g+t of a 32-block <- hidden 4g+t (t < 4) / 16+4g+t-4 (t >= 4)).  No LDS round trip, no shuffle;
//   * weights stream through one LDS buffer per matrix in chunks of HS hidden units, next chunk prefetched into
//     registers during the MFMAs of the current one (L2-resident: 147 KB ... 2.4 MB per layer);
//   * epilogue: + bias + residual in fp32, optional fused LayerNorm of the result (row statistics by two wave shuffles).
// Algorithmic HBM bytes per token: C * (4 + 2 + 4 [+ 2]) vs ~40 C in the unfused form.  MFMA work 16 M C^2.
#include "common.h"

struct SwinMlpParams {
  const float* x; const half_t* delta;
  const half_t* g2; const half_t* be2;
  const half_t* w1; const half_t* b1; const half_t* w2p; const half_t* b2;
  float* out;
  const half_t* gn; const half_t* bn; half_t* y;
  long M; float eps, eps_n;
};

// exact-GELU: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 rounding of the result)
__device__ __forceinline__ float gelu_erf(float v) {
  const float z = fabsf(v) * 0.70710678118654752f;
  const float t = __frcp_rn(1.f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = 1.f - poly * __expf(-z * z);                 // erf(|v| / sqrt(2))
  return 0.5f * v * (1.f + (v < 0.f ? -e : e));
}

template <int C, int HS>
__global__ __launch_bounds__(512) void swin_mlp_kernel(SwinMlpParams p) {
  constexpr int NW = 8, BM = 16 * NW, HID = 4 * C, KS = C / 32, CT = C / 16, NCHUNK = HID / HS, NB = HS / 32;
  constexpr int XP = C + 8, W1P = C + 8, W2P = HS + 8;        // LDS row pitches in halfs (+16 B: conflict-free b128 rows)
  constexpr int W1_PIECES = HS * (C / 8), W2_PIECES = C * (HS / 8), PIECES = W1_PIECES + W2_PIECES;
  constexpr int PPT = (PIECES + 511) / 512;                   // 16-byte pieces per thread and chunk
  static_assert(C % 32 == 0 && HS % 32 == 0 && HID % HS == 0, "tile shapes");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* xs = (half_t*)smem;                                  // [BM][XP]  prologue only; aliased by the weight chunks
  half_t* w1s = (half_t*)smem;                                 // [HS][W1P]
  half_t* w2s = w1s + HS * W1P;                                // [C][W2P]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const long row0 = (long)blockIdx.x * BM + wave * 16;        // this wave's 16 tokens

  // ---- weight chunk prefetch (global -> registers), chunk 0 goes out before the LayerNorm prologue
  half8 wreg[PPT];
  auto fetch = [&](int j) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int pc = tid + i * 512;
      if (pc < W1_PIECES) {
        const int r = pc / (C / 8), c8 = pc % (C / 8);
        wreg[i] = *(const half8*)(p.w1 + (long)(j * HS + r) * C + c8 * 8);
      } else if (pc < PIECES) {
        const int q = pc - W1_PIECES, r = q / (HS / 8), c8 = q % (HS / 8);
        wreg[i] = *(const half8*)(p.w2p + (long)r * HID + j * HS + c8 * 8);
      }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int pc = tid + i * 512;
      if (pc < W1_PIECES) {
        const int r = pc / (C / 8), c8 = pc % (C / 8);
        *(half8*)(w1s + r * W1P + c8 * 8) = wreg[i];
      } else if (pc < PIECES) {
        const int q = pc - W1_PIECES, r = q / (HS / 8), c8 = q % (HS / 8);
        *(half8*)(w2s + r * W2P + c8 * 8) = wreg[i];
      }
    }
  };
  fetch(0);

  // ---- prologue: LayerNorm of this wave's 16 rows (fp32 statistics, two-pass in registers).  LPR lanes own one row,
  // 4 consecutive channels per lane and step; rows of a wave are processed 64 / LPR at a time.
  {
    constexpr int LPR = (C / 4 <= 32) ? 32 : 64, RPP = 64 / LPR, NCH = (C / 4 + LPR - 1) / LPR;
    const int sub = lane % LPR, rsel = lane / LPR;
    half_t* xw = xs + wave * 16 * XP;
#pragma unroll 2
    for (int rr = 0; rr < 16; rr += RPP) {
      const int r = rr + rsel;
      const long row = row0 + r;
      float v[NCH][4];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int c4 = sub + k * LPR;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[k][j] = 0.f;
        if (c4 < C / 4 && row < p.M) {
          const float4_ a = *(const float4_*)(p.x + row * C + c4 * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[k][j] = a[j];
          if (p.delta) {
            const half4 d = *(const half4*)(p.delta + row * C + c4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[k][j] += (float)d[j];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) s += v[k][j];
      }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
      const float mean = s * (1.f / (float)C);
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (sub + k * LPR < C / 4) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float d = v[k][j] - mean; q += d * d; }
        }
      }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
      const float rstd = rsqrtf(q * (1.f / (float)C) + p.eps);
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int c4 = sub + k * LPR;
        if (c4 < C / 4) {
          const half4 gm = *(const half4*)(p.g2 + c4 * 4), bt = *(const half4*)(p.be2 + c4 * 4);
          half4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (half_t)((v[k][j] - mean) * rstd * (float)gm[j] + (float)bt[j]);
          *(half4*)(xw + r * XP + c4 * 4) = o;
        }
      }
    }
  }
  wave_lds_fence();
  // B-fragments of LN(x)^T: lane (g, token l15) holds x_ln[token][32 ks + 8 g .. + 7]
  half8 xf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) xf[ks] = *(const half8*)(xs + (wave * 16 + l15) * XP + ks * 32 + g * 8);

  float4_ acc2[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) acc2[ct] = (float4_){0.f, 0.f, 0.f, 0.f};

  __syncthreads();                                             // every wave has its fragments: xs may be overwritten
  stash();
  __syncthreads();

  for (int j = 0; j < NCHUNK; ++j) {
    if (j + 1 < NCHUNK) fetch(j + 1);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      // GEMM 1 (transposed): H^T[32 hidden, 16 tokens] over K = C
      float4_ h0 = (float4_){0.f, 0.f, 0.f, 0.f}, h1 = h0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const half8 a0 = *(const half8*)(w1s + (nb * 32 + l15) * W1P + ks * 32 + g * 8);
        const half8 a1 = *(const half8*)(w1s + (nb * 32 + 16 + l15) * W1P + ks * 32 + g * 8);
        h0 = mfma16(a0, xf[ks], h0);
        h1 = mfma16(a1, xf[ks], h1);
      }
      // + bias, exact GELU, fp16: the lane's 8 values are k-slots 8g .. 8g+7 of the second product
      const int hb = j * HS + nb * 32 + 4 * g;
      const half4 bb0 = *(const half4*)(p.b1 + hb), bb1 = *(const half4*)(p.b1 + hb + 16);
      half8 hf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hf[r] = (half_t)gelu_erf(h0[r] + (float)bb0[r]);
        hf[4 + r] = (half_t)gelu_erf(h1[r] + (float)bb1[r]);
      }
      // GEMM 2 (transposed): OUT^T[C, 16 tokens] += W2p[:, 32 k-slots] . H^T
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const half8 a = *(const half8*)(w2s + (ct * 16 + l15) * W2P + nb * 32 + g * 8);
        acc2[ct] = mfma16(a, hf, acc2[ct]);
      }
    }
    __syncthreads();
    if (j + 1 < NCHUNK) {
      stash();
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds OUT^T[c = 16 ct + 4 g + r][token = l15]; + bias + residual (x' re-read: L2-hot), fp32 out
  const long row = row0 + l15;
  const bool live = row < p.M;
  float s = 0.f;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int c = ct * 16 + 4 * g;
    const half4 b2 = *(const half4*)(p.b2 + c);
    float4_ xr = (float4_){0.f, 0.f, 0.f, 0.f};
    if (live) {
      xr = *(const float4_*)(p.x + row * C + c);
      if (p.delta) {
        const half4 d = *(const half4*)(p.delta + row * C + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[r] += (float)d[r];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc2[ct][r] += (float)b2[r] + xr[r]; s += acc2[ct][r]; }
    if (live) *(float4_*)(p.out + row * C + c) = acc2[ct];
  }
  if (p.y) {                                                   // fused LayerNorm of the result (next norm1 / stage norm)
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s * (1.f / (float)C);
    float q = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc2[ct][r] - mean; q += d * d; }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    const float rstd = rsqrtf(q * (1.f / (float)C) + p.eps_n);
    if (live) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int c = ct * 16 + 4 * g;
        const half4 gm = *(const half4*)(p.gn + c), bt = *(const half4*)(p.bn + c);
        half4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)((acc2[ct][r] - mean) * rstd * (float)gm[r] + (float)bt[r]);
        *(half4*)(p.y + row * C + c) = o;
      }
    }
  }
}

template <int C, int HS>
static int launch_swin_mlp(const SwinMlpParams& p, hipStream_t s) {
  constexpr int XP = C + 8, W1P = C + 8, W2P = HS + 8;
  constexpr size_t xs_b = (size_t)128 * XP * 2, w_b = (size_t)(HS * W1P + C * W2P) * 2;
  constexpr size_t smem = xs_b > w_b ? xs_b : w_b;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)swin_mlp_kernel<C, HS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const unsigned grid = (unsigned)((p.M + 127) / 128);
  hipLaunchKernelGGL((swin_mlp_kernel<C, HS>), dim3(grid), dim3(512), smem, s, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

// x [M, C] fp32, delta [M, C] fp16 or NULL, LN gamma / beta [C] fp16, w1 [4C, C], b1 [4C], w2p [C, 4C] (k-slots permuted: within every
// block of 32 hidden units slot 8g + t holds hidden unit 4g + t for t < 4 and 16 + 4g + (t - 4) for t >= 4), b2 [C] fp16 -> out [M, C] fp32 (may alias x), y [M, C] fp16 = LayerNorm(out; gn, bn, eps_n) if y != NULL.
extern "C" int mq_swin_mlp_fwd(const float* x, const void* delta, const void* ln_g, const void* ln_b, float eps, const void* w1,
                               const void* b1, const void* w2p, const void* b2, float* out, const void* next_g, const void* next_b,
                               float eps_next, void* y, long M, int C, void* stream) {
  if (M <= 0) return 0;
  SwinMlpParams p;
  p.x = x; p.delta = (const half_t*)delta; p.g2 = (const half_t*)ln_g; p.be2 = (const half_t*)ln_b; p.eps = eps;
  p.w1 = (const half_t*)w1; p.b1 = (const half_t*)b1; p.w2p = (const half_t*)w2p; p.b2 = (const half_t*)b2; p.out = out;
  p.gn = (const half_t*)next_g; p.bn = (const half_t*)next_b; p.eps_n = eps_next; p.y = (half_t*)y; p.M = M;
  if (y && (!next_g || !next_b)) return -2;
  hipStream_t s = (hipStream_t)stream;
  switch (C) {
    case 96: return launch_swin_mlp<96, 64>(p, s);
    case 192: return launch_swin_mlp<192, 32>(p, s);
    case 384: return launch_swin_mlp<384, 32>(p, s);
    default: return -1;                                      // other widths: library GEMM path of the caller
  }
}
