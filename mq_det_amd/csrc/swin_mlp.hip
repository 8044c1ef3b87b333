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
//   * a workgroup owns 128 T tokens (8 waves x 16 T); the residual stream x is fp32 in HBM (read once, written once);
//   * LayerNorm is the prologue and needs no LDS: a lane loads its token's channels directly in MFMA B-fragment order
//     (the 4 lanes of a token read one full 128-byte line per 32 channels), row statistics are an in-lane sum plus two
//     shuffles, and the normalised fp16 fragments stay in registers for the whole kernel;
//   * both GEMMs are computed TRANSPOSED:  H^T = W1 . LN(x)^T  and  OUT^T = W2 . H^T.  With v_mfma_f32_16x16x32_f16 the
//     accumulator of the first product (lane holds 4 consecutive hidden units of one token) is, after bias + GELU + cvt,
//     exactly a B-fragment of the second one -- provided the k-slots of W2 are permuted accordingly (done once on the
//     host: slot 8g+t of a 32-block <- hidden 4g+t (t < 4) / 16+4g+t-4 (t >= 4)).  No LDS round trip, no shuffle;
//   * weights stream through a double-buffered LDS chunk of HS hidden units (one barrier per chunk), the next chunk is
//     prefetched into registers during the MFMAs of the current one (L2-resident: 147 KB ... 2.4 MB per layer); their
//     A-fragments go through a 3-deep register ring so that LDS latency hides under the preceding MFMAs, and each is
//     re-used for T = 2 token blocks where the register file allows (C <= 192);
//   * epilogue: + bias + residual in fp32, optional fused LayerNorm of the result (row statistics by two wave shuffles).
// Algorithmic HBM bytes per token: C * (4 + 2 + 4 [+ 2]) vs ~40 C in the unfused form.  MFMA work 16 M C^2.
#include "common.h"
#include <type_traits>
#include <cstdlib>

MQ_NAMESPACE_BEGIN

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

// T = 16-token blocks per wave (A-fragments of the weights are re-used T times: LDS traffic per MFMA / T)
// NW = waves per workgroup (tokens per workgroup = 16 T NW).  Small workgroups (NW = 4) let several INDEPENDENT workgroups share
// a CU, so that one's memory-bound prologue / epilogue and GELU arithmetic overlap another's MFMA phase; within one
// workgroup the waves run in lock step between the per-chunk barriers.
template <int C, int HS, int T, int NW>
__global__ __launch_bounds__(64 * NW) void swin_mlp_kernel(SwinMlpParams p) {
  constexpr int NT = 64 * NW, TW = 16 * T, BM = TW * NW, HID = 4 * C, KS = C / 32, CT = C / 16, NCHUNK = HID / HS, NB = HS / 32;
  constexpr int W1P = C + 8, W2P = HS + 8;                    // LDS row pitches in halfs (+16 B: conflict-free b128 rows)
  constexpr int CHUNK = HS * W1P + C * W2P;                   // halfs per weight chunk (W1 rows | W2 rows)
  constexpr int W1_PIECES = HS * (C / 8), W2_PIECES = C * (HS / 8), PIECES = W1_PIECES + W2_PIECES;
  constexpr int PPT = (PIECES + NT - 1) / NT;                  // 16-byte pieces per thread and chunk
  static_assert(C % 32 == 0 && HS % 32 == 0 && HID % HS == 0, "tile shapes");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* wbuf = (half_t*)smem;                                // [2][CHUNK]: double-buffered weight chunks

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const long row0 = (long)blockIdx.x * BM + wave * TW;        // this wave's 16 * T tokens

  // ---- weight chunk pipeline: global -> registers (issued one chunk ahead) -> LDS buffer (j & 1)
  half8 wreg[PPT];
  auto fetch = [&](int j) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int pc = tid + i * NT;
      if (pc < W1_PIECES) {
        const int r = pc / (C / 8), c8 = pc % (C / 8);
        wreg[i] = *(const half8*)(p.w1 + (long)(j * HS + r) * C + c8 * 8);
      } else if (pc < PIECES) {
        const int q = pc - W1_PIECES, r = q / (HS / 8), c8 = q % (HS / 8);
        wreg[i] = *(const half8*)(p.w2p + (long)r * HID + j * HS + c8 * 8);
      }
    }
  };
  auto stash = [&](int buf) {
    half_t* w1s = wbuf + buf * CHUNK;
    half_t* w2s = w1s + HS * W1P;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int pc = tid + i * NT;
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

  // ---- prologue: LayerNorm straight into MFMA B-fragments.  Lane (g, token l15) loads x[token][32 ks + 8 g .. + 7] -- the
  // four g-lanes of a token cover 128 contiguous bytes per ks, a full cache line -- so a row lives in 4 lanes: statistics
  // are an in-lane sum plus two shuffles, and the normalised values already sit where the first MFMA wants them.
  // All loads of a token block are issued before anything consumes them: rows beyond M are clamped (their results are never
  // stored) and the optional delta is a wave-uniform choice made OUTSIDE the loop -- with `if (live)` / `if (p.delta)` inside it
  // every k-step was its own basic block ending in s_waitcnt vmcnt(0): KS serialised HBM round trips per wave (ISA of v2).
  half8 xf[T][KS];
  auto prologue = [&](auto HAS_DELTA) {
    constexpr bool has_delta = decltype(HAS_DELTA)::value;
#pragma unroll
    for (int tb = 0; tb < T; ++tb) {
      const long row = min(row0 + tb * 16 + l15, p.M - 1);
      const float* xr = p.x + row * C + g * 8;
      float4_ va[KS], vb[KS];
      half8 vd[has_delta ? KS : 1];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        va[ks] = *(const float4_*)(xr + ks * 32);
        vb[ks] = *(const float4_*)(xr + ks * 32 + 4);
        if constexpr (has_delta) vd[ks] = *(const half8*)(p.delta + row * C + ks * 32 + g * 8);
      }
      float v[KS][8];
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[ks][j] = va[ks][j]; v[ks][4 + j] = vb[ks][j]; }
        if constexpr (has_delta) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[ks][j] += (float)vd[ks][j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[ks][j];
      }
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const float mean = s * (1.f / (float)C);
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[ks][j] - mean; q += d * d; }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      const float rstd = rsqrtf(q * (1.f / (float)C) + p.eps);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c = ks * 32 + g * 8;
        const half8 gm = *(const half8*)(p.g2 + c), bt = *(const half8*)(p.be2 + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[tb][ks][j] = (half_t)((v[ks][j] - mean) * rstd * (float)gm[j] + (float)bt[j]);
      }
    }
  };
  if (p.delta) prologue(std::true_type{}); else prologue(std::false_type{});

  float4_ acc2[T][CT];
#pragma unroll
  for (int tb = 0; tb < T; ++tb)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc2[tb][ct] = (float4_){0.f, 0.f, 0.f, 0.f};

  stash(0);
  __syncthreads();

  for (int j = 0; j < NCHUNK; ++j) {
    if (j + 1 < NCHUNK) fetch(j + 1);
    const half_t* w1s = wbuf + (j & 1) * CHUNK;
    const half_t* w2s = w1s + HS * W1P;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      // GEMM 1 (transposed): H^T[32 hidden, 16 T tokens] over K = C; A-fragments through a 3-deep register ring so that the
      // LDS latency of step ks + 2 hides under the MFMAs of steps ks, ks + 1
      float4_ h0[T], h1[T];
#pragma unroll
      for (int tb = 0; tb < T; ++tb) h0[tb] = h1[tb] = (float4_){0.f, 0.f, 0.f, 0.f};
      const half_t* a0p = w1s + (nb * 32 + l15) * W1P + g * 8;
      const half_t* a1p = a0p + 16 * W1P;
      half8 r0[3], r1[3];
#pragma unroll
      for (int ks = 0; ks < 2 && ks < KS; ++ks) { r0[ks] = *(const half8*)(a0p + ks * 32); r1[ks] = *(const half8*)(a1p + ks * 32); }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 2 < KS) { r0[(ks + 2) % 3] = *(const half8*)(a0p + (ks + 2) * 32); r1[(ks + 2) % 3] = *(const half8*)(a1p + (ks + 2) * 32); }
#pragma unroll
        for (int tb = 0; tb < T; ++tb) {
          h0[tb] = mfma16(r0[ks % 3], xf[tb][ks], h0[tb]);
          h1[tb] = mfma16(r1[ks % 3], xf[tb][ks], h1[tb]);
        }
      }
      // first fragments of the second product go out before the GELU arithmetic
      const half_t* a2p = w2s + l15 * W2P + nb * 32 + g * 8;
      half8 r2[3];
#pragma unroll
      for (int ct = 0; ct < 2 && ct < CT; ++ct) r2[ct] = *(const half8*)(a2p + ct * 16 * W2P);
      // + bias, exact GELU, fp16: the lane's 8 values are k-slots 8g .. 8g+7 of the second product
      const int hb = j * HS + nb * 32 + 4 * g;
      const half4 bb0 = *(const half4*)(p.b1 + hb), bb1 = *(const half4*)(p.b1 + hb + 16);
      half8 hf[T];
#pragma unroll
      for (int tb = 0; tb < T; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          hf[tb][r] = (half_t)gelu_erf(h0[tb][r] + (float)bb0[r]);
          hf[tb][4 + r] = (half_t)gelu_erf(h1[tb][r] + (float)bb1[r]);
        }
      // GEMM 2 (transposed): OUT^T[C, 16 T tokens] += W2p[:, 32 k-slots] . H^T
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        if (ct + 2 < CT) r2[(ct + 2) % 3] = *(const half8*)(a2p + (ct + 2) * 16 * W2P);
#pragma unroll
        for (int tb = 0; tb < T; ++tb) acc2[tb][ct] = mfma16(r2[ct % 3], hf[tb], acc2[tb][ct]);
      }
    }
    if (j + 1 < NCHUNK) stash((j + 1) & 1);
    __syncthreads();                                           // chunk j + 1 visible; buffer (j & 1) free for chunk j + 2
  }

  // ---- epilogue: lane holds OUT^T[c = 16 ct + 4 g + r][token = l15]; + bias + residual (x' re-read: L2-hot), fp32 out
#pragma unroll
  for (int tb = 0; tb < T; ++tb) {
    const long row = row0 + tb * 16 + l15;
    const bool live = row < p.M;
    float s = 0.f;
    // residual x' = x (+ delta), re-read L2-hot, in groups of EG channel blocks: the loads of a group are all in flight before
    // the first add / store (one load -> wait -> store chain per block before: `out` may alias `x`, so the compiler could not
    // move a load above the previous store -- a lane only ever re-reads the addresses it writes itself, which makes it safe here)
    constexpr int EG = 6;                                  // CT = C / 16 = 6, 12, 24
    static_assert(CT % EG == 0, "channel blocks per group");
    const long rrow = min(row, p.M - 1);
#pragma unroll
    for (int ct0 = 0; ct0 < CT; ct0 += EG) {
      float4_ xr[EG];
      half4 dl[EG], b2[EG];
#pragma unroll
      for (int i = 0; i < EG; ++i) {
        const int c = (ct0 + i) * 16 + 4 * g;
        xr[i] = *(const float4_*)(p.x + rrow * C + c);
        b2[i] = *(const half4*)(p.b2 + c);
      }
      if (p.delta) {
#pragma unroll
        for (int i = 0; i < EG; ++i) dl[i] = *(const half4*)(p.delta + rrow * C + (ct0 + i) * 16 + 4 * g);
      } else {
#pragma unroll
        for (int i = 0; i < EG; ++i) dl[i] = (half4){(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
      }
#pragma unroll
      for (int i = 0; i < EG; ++i) {
        const int ct = ct0 + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc2[tb][ct][r] += (float)b2[i][r] + (xr[i][r] + (float)dl[i][r]); s += acc2[tb][ct][r]; }
      }
#pragma unroll
      for (int i = 0; i < EG; ++i)
        if (live) *(float4_*)(p.out + row * C + (ct0 + i) * 16 + 4 * g) = acc2[tb][ct0 + i];
    }
    if (p.y) {                                                 // fused LayerNorm of the result (next norm1 / stage norm)
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const float mean = s * (1.f / (float)C);
      float q = 0.f;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = acc2[tb][ct][r] - mean; q += d * d; }
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
          for (int r = 0; r < 4; ++r) o[r] = (half_t)((acc2[tb][ct][r] - mean) * rstd * (float)gm[r] + (float)bt[r]);
          *(half4*)(p.y + row * C + c) = o;
        }
      }
    }
  }
}

template <int C, int HS, int T, int NW>
static int launch_swin_mlp(const SwinMlpParams& p, hipStream_t s) {
  constexpr size_t smem = (size_t)2 * (HS * (C + 8) + C * (HS + 8)) * 2;
  static MqOncePerDevice attr;
  if (attr.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)swin_mlp_kernel<C, HS, T, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr.done();
  }
  constexpr int BM = 16 * T * NW;
  const unsigned grid = (unsigned)((p.M + BM - 1) / BM);
  hipLaunchKernelGGL((swin_mlp_kernel<C, HS, T, NW>), dim3(grid), dim3(64 * NW), smem, s, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

// x [M, C] fp32, delta [M, C] fp16 or NULL, LN gamma / beta [C] fp16, w1 [4C, C], b1 [4C], w2p [C, 4C] (k-slots permuted: within every
// block of 32 hidden units slot 8g + t holds hidden unit 4g + t for t < 4 and 16 + 4g + (t - 4) for t >= 4), b2 [C] fp16 -> out [M, C] fp32 (may alias x), y [M, C] fp16 = LayerNorm(out; gn, bn, eps_n) if y != NULL.
extern "C" int MQ_SYM(mq_swin_mlp_fwd)(const float* x, const void* delta, const void* ln_g, const void* ln_b, float eps, const void* w1,
                               const void* b1, const void* w2p, const void* b2, float* out, const void* next_g, const void* next_b,
                               float eps_next, void* y, long M, int C, void* stream) {
  if (M <= 0) return 0;
  SwinMlpParams p;
  p.x = x; p.delta = (const half_t*)delta; p.g2 = (const half_t*)ln_g; p.be2 = (const half_t*)ln_b; p.eps = eps;
  p.w1 = (const half_t*)w1; p.b1 = (const half_t*)b1; p.w2p = (const half_t*)w2p; p.b2 = (const half_t*)b2; p.out = out;
  p.gn = (const half_t*)next_g; p.bn = (const half_t*)next_b; p.eps_n = eps_next; p.y = (half_t*)y; p.M = M;
  if (y && (!next_g || !next_b)) return -2;
  hipStream_t s = (hipStream_t)stream;
  // tile shapes: the fastest measured per width in round 2 (profiles/README.md; the slower ones -- 64-token chunks, T = 2, 4 waves at
  // C = 384 -- are gone with their A/B switch)
  switch (C) {
    case 96: return launch_swin_mlp<96, 32, 1, 4>(p, s);
    case 192: return launch_swin_mlp<192, 32, 1, 4>(p, s);
    case 384: return launch_swin_mlp<384, 32, 1, 8>(p, s);  // 8 waves: 0.33 ms per launch vs 0.38 with 4
    default: return -1;                                      // other widths: library GEMM path of the caller
  }
}

MQ_NAMESPACE_END
