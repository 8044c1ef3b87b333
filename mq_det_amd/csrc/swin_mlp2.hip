// mq_swin_mlp2_fwd: the MLP half of a Swin block as ONE kernel for gfx950 -- second generation (round 3).
//
//     x'   = x + delta                       (delta = attention projection output, swint.py:236; optional)
//     out  = x' + fc2( GELU( fc1( LayerNorm(x') ) ) )        (swint.py:240, Mlp :13-31, exact erf GELU)
//     y    = LayerNorm_next(out)             (optional: the next block's norm1 / the stage's output norm, 16-bit)
//
// Same mathematics and register-chained transposed GEMM pair as the first-generation kernel (swin_mlp.hip, removed in round 5: H^T = W1 LN(x)^T,
// OUT^T = W2 H^T; the accumulator of the first product IS the B fragment of the second).  What round 3's counters said about that kernel (profiles/r03_call1_swin_mlp_sq*.csv,
// MI355X): MFMA pipe 9 % busy; 6.4 / 12.6 / 24 VALU instructions per MFMA at C = 384 / 192 / 96 -- the exact GELU was ~23 VALU per
// value (__frcp_rn expands to the IEEE division sequence); 46 % of the LDS cycles were bank conflicts (padded row-major weight
// chunks); per 32-hidden chunk a barrier, with GEMM1 -> GELU -> GEMM2 strictly one after the other inside every wave, so the two
// waves of a SIMD were always in the same phase (both on the matrix pipe, then both on the VALU).  This kernel changes four things:
//
//   1. weights arrive FRAGMENT-MAJOR (packed once on the host, ops.swin_mlp2_pack): every 1 KB block is one MFMA A fragment in lane
//      order.  Staging is a linear LDS-DMA copy (global_load_lds_dwordx4: no VGPRs, no ds_write), fragment reads are
//      `base + lane * 16 + immediate` -- conflict-free by construction, no address arithmetic;
//   2. a three-deep SOFTWARE PIPELINE over the hidden chunks inside each wave: iteration j issues GELU(chunk j) [VALU] together with
//      GEMM1(chunk j + 1) and GEMM2(chunk j - 1) [MFMA], all mutually independent, so the matrix pipe and the VALU overlap within
//      one instruction stream (W1 runs two chunks ahead of W2 through two small LDS rings; one barrier per iteration);
//   3. a GELU of 14 VALU instructions per value (v_rcp_f32 + v_exp_f32 + v_bfi, fc1 bias folded into the accumulator's initial
//      value), or 7 + one ds_read_b64 with the interpolation table variant (768 x (Phi, dPhi) over [-6, 6), |error| < 8e-6);
//   4. nothing in the loop is conditional: the packed W1 carries two zero chunks behind the last one.
// Algorithmic HBM bytes per token: C * (4 + 2 + 4 [+ 2]).  MFMA work 16 M C^2.
#include "common.h"
#include <type_traits>
#include <cstdlib>

MQ_NAMESPACE_BEGIN

struct SwinMlp2Params {
  const float* x; const half_t* delta;
  const half_t* g2; const half_t* be2;
  const half_t* w1f; const half_t* b1; const half_t* w2f; const half_t* b2;
  float* out;
  const half_t* gn; const half_t* bn; half_t* y;
  long M; float eps, eps_n;
};

// SPLIT-PRECISE build (-DMQ_F32, round 6).  Every weight fragment feeds exactly ONE MFMA per wave here, so splitting operands inside mfma16 (round 6's
// first step) costs ~40 VALU instructions per three MFMAs -- VALU-bound at a third of the matrix rate.  Instead the host packs the weights ALREADY
// SPLIT (ops.swin_mlp2_pack under the precise mode): every 512-element fragment block is [hi: 64 lanes x 8 fp16 | lo: 64 lanes x 8 fp16] -- the same
// 2 KB as the fp32 block -- staged by two linear LDS-DMA pieces and read back as one ds_read_b128 per plane; the LayerNorm fragments are split once
// per wave, the GELU output once per hidden chunk.  No conversion arithmetic is left beside the weight stream.
#if defined(MQ_F32) && !defined(MQ_F32_EXACT)
#define MQ_SW_SPLIT 1
typedef mq_split8 wfrag_t;
__device__ __forceinline__ wfrag_t sw_frag(half8 x) { return mq_split(x); }
__device__ __forceinline__ float4_ sw_mfma(const wfrag_t& a, const wfrag_t& b, float4_ c) { return mfma16_split(a, b, c); }
// fragment of lane `lane` out of a planar block (LDS or global): hi at fp16 index 8 lane, lo 512 fp16 further
__device__ __forceinline__ wfrag_t sw_wfrag(const half_t* blk, int lane) {
  const _Float16* h = (const _Float16*)blk + lane * 8;
  wfrag_t f;
  f.hi = *(const mq_h16x8*)h;
  f.lo = *(const mq_h16x8*)(h + 512);
  return f;
}
#define SW_LANE8 0                       // the lane offset is applied inside sw_wfrag (fp16 units of a plane)
#define SW_WFRAG(ptr) sw_wfrag((ptr), lane)
#else
#define MQ_SW_SPLIT 0
typedef half8 wfrag_t;
#define sw_frag(x) (x)
#define sw_mfma(a, b, c) mfma16((a), (b), (c))
#define SW_LANE8 (lane * 8)              // 16-bit builds: token for token the round-5 expressions (tools/isa_diff.py: identical ISA)
#define SW_WFRAG(ptr) (*(const half8*)(ptr))
#endif

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): erf(|v| / sqrt 2) = 1 - poly(t) exp(-v^2 / 2), t = 1 / (1 + p |v| / sqrt 2)
__device__ __forceinline__ float erf_abs_scaled(float v) {
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, fabsf(v), 1.f));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float ex = __builtin_amdgcn_exp2f((-0.5f * 1.4426950408889634f) * v * v);
  return fmaf(-poly, ex, 1.f);
}
// exact-erf GELU, 14 VALU instructions (2 of them transcendental)
__device__ __forceinline__ float gelu_erf2(float v) {
  const float hv = 0.5f * v;
  return fmaf(hv, __builtin_copysignf(erf_abs_scaled(v), v), hv);
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <int I, int N, class F>
__device__ __forceinline__ void static_for_impl(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_impl<I + 1, N>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void static_for(F f) { static_for_impl<0, N>(f); }

// GELU piece q of an iteration (8 values x GS stages in stage-diagonal order: value k = d - stage on diagonal d, so that a value's
// stages are a few steps apart): its value index (want_k) or its stage
constexpr int gelu_piece_of(int q, int GS, bool want_k) {
  int c = 0;
  for (int d = 0; d < 8 + GS - 1; ++d)
    for (int st = 0; st < GS; ++st) {
      const int k = d - st;
      if (k < 0 || k >= 8) continue;
      if (c == q) return want_k ? k : st;
      ++c;
    }
  return -1;
}
#define MQ_GELU_TAB_N 768            // Phi on [-6, 6) in steps of 1 / 64: linear interpolation error <= h^2 / 8 max|Phi''| = 7.4e-6

// LayerNorm of the wave's 16 tokens straight into MFMA B fragments (lane (g, token l15) loads x[token][32 ks + 8 g .. + 7],
// the four g-lanes of a token cover one 128-byte line per ks; statistics = in-lane sum + two shuffles; rows beyond M are clamped)
template <int C, bool HAS_DELTA>
__device__ __forceinline__ void ln_fragments(const SwinMlp2Params& p, long row0, int l15, int g, half8 (&xf)[C / 32]) {
  constexpr int KS = C / 32;
  const long row = min(row0 + l15, p.M - 1);
  const float* xr = p.x + row * C + g * 8;
  float4_ va[KS], vb[KS];
  half8 vd[HAS_DELTA ? KS : 1];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    va[ks] = *(const float4_*)(xr + ks * 32);
    vb[ks] = *(const float4_*)(xr + ks * 32 + 4);
    if constexpr (HAS_DELTA) vd[ks] = *(const half8*)(p.delta + row * C + ks * 32 + g * 8);
  }
  float v[KS][8];
  float s = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[ks][j] = va[ks][j]; v[ks][4 + j] = vb[ks][j]; }
    if constexpr (HAS_DELTA) {
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
    for (int j = 0; j < 8; ++j) xf[ks][j] = (half_t)((v[ks][j] - mean) * rstd * (float)gm[j] + (float)bt[j]);
  }
}

// Weights travel D = 1 iteration ahead of their use through two-stage LDS rings.  (A D = 2 / three-stage variant, with a counted
// s_waitcnt vmcnt that kept the youngest pieces in flight across the barrier, was built and measured in round 3 -- GPU calls 2 / 3,
// profiles/r03_call3_microbench_swin_mlp.json: 0.2207 vs 0.2199 ms at C = 384, 0.206 vs 0.189 at C = 192 (one wave per SIMD less) --
// the DMA latency is not what bounds an iteration; removed.)
template <int C, int NW, bool TABLE>
// (split-precise: the planar fragments are twice the registers and the weight rings twice the LDS -- C = 96: two workgroups per SIMD set, C = 192:
// one, which the 100 KB of its rings allow anyway)
__global__ __launch_bounds__(64 * NW, MQ_SW_SPLIT ? (C <= 96 ? 2 : 1) : (C <= 96 ? 4 : C <= 192 ? 3 : 2)) void swin_mlp2_kernel(SwinMlp2Params p) {
  static_assert(!(MQ_SW_SPLIT && C >= 384) || NW == 4, "split-precise C = 384: four waves (one per SIMD)");
  constexpr int NT = 64 * NW, BM = 16 * NW, HID = 4 * C, KS = C / 32, CT = C / 16, NCHUNK = HID / 32;
  constexpr int FR = 512;                                     // halfs per fragment block (64 lanes x 8)
  constexpr int W1_FR = 2 * KS, W2_FR = CT, IT_FR = W1_FR + W2_FR;   // fragment blocks of one chunk of W1 / W2 / staged per iteration
  constexpr int FPW = IT_FR / NW;                             // fragment blocks a wave stages per iteration
  constexpr int D = 1, NS = D + 1, U = 2;                     // prefetch distance, ring stages, unroll = lcm(2, NS)
  // W2ONE (split-precise, C = 384): two stages of both rings would be 192 KB of planar fragments.  W1 keeps its two stages (96 KB); W2 has ONE
  // (48 KB): chunk j is loaded at the END of iteration j -- behind the barrier that says everybody has read chunk j - 1 -- and waited for before
  // iteration j + 1 starts: one exposed L2 -> LDS copy of 48 KB per iteration, against streaming ALL weights from L2 for every 16 tokens (the
  // tail kernel, which is what round 5's precise mode ran this width through: 1.04 ms per launch).
  constexpr bool W2ONE = MQ_SW_SPLIT && C >= 384;
  constexpr int NS2 = W2ONE ? 1 : NS;
  static_assert(C % 32 == 0 && IT_FR % NW == 0 && NCHUNK % U == 0, "tile shapes");
  static_assert(!W2ONE || (W1_FR % NW == 0 && W2_FR % NW == 0), "per-ring staging");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* w1s = (half_t*)smem;                                // [NS][W1_FR][FR]   W1 ring: chunk c lives in stage c % NS
  half_t* w2s = w1s + NS * W1_FR * FR;                        // [NS2][W2_FR][FR]  W2 ring
  half_t* b1s = w2s + NS2 * W2_FR * FR;                       // [HID]             fc1 bias (no ordinary global load inside the loop)
  float* tab = (float*)(b1s + HID);                           // [MQ_GELU_TAB_N][2] (TABLE)

  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: staging addresses in SGPRs
  const long row0 = (long)blockIdx.x * BM + wave * 16;        // this wave's 16 tokens

  const mq_rsrc rs_w1 = mq_raw_buffer(p.w1f), rs_w2 = mq_raw_buffer(p.w2f);
  // ---- staging of one iteration's weights by LDS-DMA: W1 chunk c1 -> W1 stage s1, W2 chunk c2 -> W2 stage s2.  Block f of the iteration
  // (f < W1_FR: W1, else W2) is copied by wave f % NW as one 1 KB piece: source and destination are both lane-linear.
  auto stage_issue = [&](int c1, int s1, int c2, int s2) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
      const int f = wave + i * NW;
      // (MUBUF `buffer_load ... lds`, csrc/common.h: with the FLAT-encoded global_load_lds every wait of the loop below was lgkmcnt(0))
      const mq_rsrc rs = f < W1_FR ? rs_w1 : rs_w2;
      const int eoff = f < W1_FR ? (c1 * W1_FR + f) * FR : (c2 * W2_FR + (f - W1_FR)) * FR;
      half_t* dst = f < W1_FR ? w1s + (s1 * W1_FR + f) * FR : w2s + (s2 * W2_FR + (f - W1_FR)) * FR;
#if MQ_SW_SPLIT
      // the 2 KB planar block as two linear 1 KB LDS-DMA pieces (hi plane, lo plane): asynchronous like the 16-bit builds' one piece
      lds_stage_16b_buf(rs, eoff * (int)sizeof(half_t), lane * 16, dst);
      lds_stage_16b_buf(rs, eoff * (int)sizeof(half_t) + 1024, lane * 16, (char*)dst + 1024);
#else
      lds_stage_frag8_buf(rs, nullptr, eoff, lane * 8, dst, lane);
#endif
    }
  };
  // (W2ONE) the two rings staged separately: `nb` fragment blocks from `g` to `l`, block f by wave f % NW
  auto stage_ring = [&](mq_rsrc rs, int g0, half_t* l, auto NBc) __attribute__((always_inline)) {    // g0: element offset inside the array of rs
    constexpr int nb = decltype(NBc)::value;
#pragma unroll
    for (int i = 0; i < nb / NW; ++i) {
      const int f = wave + i * NW;
#if MQ_SW_SPLIT
      char* dst = (char*)(l + f * FR);
      lds_stage_16b_buf(rs, (g0 + f * FR) * (int)sizeof(half_t), lane * 16, dst);
      lds_stage_16b_buf(rs, (g0 + f * FR) * (int)sizeof(half_t) + 1024, lane * 16, dst + 1024);
#else
      lds_stage_frag8_buf(rs, nullptr, g0 + f * FR, lane * 8, l + f * FR, lane);
#endif
    }
  };
  // chunks 0 .. D of W1 and chunk 0 of W2 -- into its own stage 0 and into stage NS - 1, which iteration 0 reads for its GEMM2 of the
  // (all-zero) H of "chunk -1": finite weights x 0 = 0, so iteration 0 needs no special case -- : everything the fill and the first
  // iterations read before their own pieces land
  if constexpr (W2ONE) {
#pragma unroll
    for (int c = 0; c <= D; ++c) stage_ring(rs_w1, c * W1_FR * FR, w1s + (c % NS) * W1_FR * FR, std::integral_constant<int, W1_FR>{});
    stage_ring(rs_w2, 0, w2s, std::integral_constant<int, W2_FR>{});
  } else {
#pragma unroll
    for (int c = 0; c <= D; ++c) stage_issue(c, c % NS, 0, c == 0 ? 0 : NS - 1);
  }
  for (int i = tid; i < HID / 8; i += NT) *(half8*)(b1s + i * 8) = *(const half8*)(p.b1 + i * 8);
  if constexpr (TABLE) {
    // Phi(x_i), Phi(x_i+1) - Phi(x_i) at x_i = -6 + i / 64
    for (int i = tid; i < MQ_GELU_TAB_N; i += NT) {
      const float x0 = -6.f + (float)i * (1.f / 64.f), x1 = x0 + (1.f / 64.f);
      const float p0 = fmaf(0.5f, __builtin_copysignf(erf_abs_scaled(x0), x0), 0.5f);
      const float p1 = fmaf(0.5f, __builtin_copysignf(erf_abs_scaled(x1), x1), 0.5f);
      tab[2 * i] = p0;
      tab[2 * i + 1] = p1 - p0;
    }
  }

#if MQ_SW_SPLIT
  wfrag_t xf[KS];
  {
    half8 xr[KS];
    if (p.delta) ln_fragments<C, true>(p, row0, l15, g, xr); else ln_fragments<C, false>(p, row0, l15, g, xr);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = sw_frag(xr[ks]);       // split once per wave
  }
#else
  half8 xf[KS];
  if (p.delta) ln_fragments<C, true>(p, row0, l15, g, xf); else ln_fragments<C, false>(p, row0, l15, g, xf);
#endif

  __syncthreads();                                            // (drains the DMAs: vmcnt(0)) W1 chunks 0 .. D, W2 chunk 0, bias, table visible

  float4_ acc2[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) acc2[ct] = (float4_){0.f, 0.f, 0.f, 0.f};

  // fc1 bias of chunk c = the initial value of its accumulators (lane holds hidden units 4 g + r and 16 + 4 g + r of the chunk); from LDS:
  // an ordinary global load inside the loop would make hipcc drain the LDS-DMA queue at its first use (VMEM returns in order)
  auto h_init = [&](int c, float4_ (&h)[2]) __attribute__((always_inline)) {
    const int hb = min(c, NCHUNK - 1) * 32 + 4 * g;
    const half4 b0 = *(const half4*)(b1s + hb), b1 = *(const half4*)(b1s + hb + 16);
#pragma unroll
    for (int r = 0; r < 4; ++r) { h[0][r] = (float)b0[r]; h[1][r] = (float)b1[r]; }
  };
  // GEMM 1 (transposed) of one chunk from W1 stage `st`: H^T[32 hidden, 16 tokens] over K = C (pipeline fill only)
  auto gemm1 = [&](int st, float4_ (&h)[2]) {
    const half_t* a = w1s + st * W1_FR * FR + SW_LANE8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      h[0] = sw_mfma(SW_WFRAG(a + ks * FR), xf[ks], h[0]);
      h[1] = sw_mfma(SW_WFRAG(a + (KS + ks) * FR), xf[ks], h[1]);
    }
  };
  // GEMM 2 (transposed) of one chunk from W2 stage `st`: OUT^T[C, 16 tokens] += W2p[:, 32 k-slots] . H^T (pipeline drain only)
  auto gemm2 = [&](int st, const half8& hf) {
    const half_t* a = w2s + st * W2_FR * FR + SW_LANE8;
#if MQ_SW_SPLIT
    const wfrag_t hs = sw_frag(hf);
#else
    const half8& hs = hf;
#endif
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc2[ct] = sw_mfma(SW_WFRAG(a + ct * FR), hs, acc2[ct]);
  };

  // ---- one pipelined iteration.  Its MFMAs form ONE sequence of N = 2 KS + CT steps that alternates GEMM1(chunk j + 1) -- a
  // dependent chain on two accumulators -- with GEMM2(chunk j - 1), whose CT accumulators are independent; their A fragments come
  // through a register ring RD reads deep (LDS latency, ~100+ cycles, against ~30 per step).  GELU(chunk j) is cut into 8 values x
  // GS stages (~5 VALU each, erf: {rcp + exp2 | polynomial | sign, scale, round}; table: {index + ds_read | interpolate}) that are
  // dealt out over the steps.  The source order IS the issue order: a scheduling fence closes every step -- left alone, the
  // compiler emits the whole GELU first and then MFMA after MFMA, each waiting for a fragment read issued one step earlier (ISA).
  constexpr int RD = 6, GS = TABLE ? 2 : 3, NPIECE = 8 * GS;
  // MQ_PIN(x): an empty volatile asm that reads and "writes" x.  Instruction selection linearises a block's DAG on its own: arithmetic
  // that hangs on no side-effecting node is placed wherever it likes relative to the scheduling fences.  A piece's input and result
  // both pass through a pin, which chains the piece between the two fences of its step.
#define MQ_PIN(x) asm volatile("" : "+v"(x))
  // (gts / ges: the iteration's GELU state between stages, local to the step -- every value's stages begin and end inside one step)
  // (table: stage 0 only ISSUES the table read -- its two words travel to stage 1 untouched.  Used in stage 0, as in round 5, the LDS read is
  // waited for with lgkmcnt(0) on the spot: LDS returns in order, so that wait also drains the RD fragment reads just issued for the coming MFMAs)
  auto gelu_piece = [&](auto kc, auto stc, const float4_ (&hin)[2], half8& hf, float (&gts)[8], float (&ges)[8], float (&gds)[8]) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value, stage = decltype(stc)::value;
    float& gt = gts[k];
    float& ge = ges[k];
    float v = hin[k >> 2][k & 3];
    MQ_PIN(v);
    if constexpr (TABLE) {
      if constexpr (stage == 0) {
        const float pos = fmaf(__builtin_amdgcn_fmed3f(v, -6.f, 5.9921875f), 64.f, 384.f);        // in [0, 768)
        const int idx = (int)pos;
        gt = pos - (float)idx;
        MQ_PIN(gt);
        const float2_ e = *(const float2_*)(tab + 2 * idx);
        ge = e[0];
        gds[k] = e[1];
      } else {
        float t = gt * gds[k];                                // frac * dPhi, rounded, + Phi_i: the arithmetic of round 5
        MQ_PIN(t);
        float o = v * (t + ge);
        MQ_PIN(o);
        hf[k] = (half_t)o;
      }
    } else {
      (void)gds;
      if constexpr (stage == 0) {
        gt = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, fabsf(v), 1.f));
        ge = __builtin_amdgcn_exp2f((-0.5f * 1.4426950408889634f) * v * v);
        MQ_PIN(gt);
        MQ_PIN(ge);
      } else if constexpr (stage == 1) {
        const float t = gt;
        gt = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
        MQ_PIN(gt);
      } else {
        const float hv = 0.5f * v;
        float o = fmaf(hv, __builtin_copysignf(fmaf(-gt, ge, 1.f), v), hv);
        MQ_PIN(o);
        hf[k] = (half_t)o;
      }
    }
  };
  auto step = [&](int j, auto PHc, auto PARc, float4_ (&hin)[2], float4_ (&hout)[2], half8& hf_old, half8& hf_new)
      __attribute__((always_inline)) {
    constexpr int PH = decltype(PHc)::value;                  // j % NS
    constexpr int N = 2 * KS + CT;
    constexpr int S1 = (PH + 1) % NS, S2 = W2ONE ? 0 : (PH + NS - 1) % NS;          // stages of W1 chunk j + 1 / W2 chunk j - 1 (read now)
    // pieces for iteration j + D: W1 chunk j + 1 + D -> the stage W1 chunk j left (read in iteration j - 1), W2 chunk j - 1 + D -> the stage
    // of W2 chunk j - 2 (ditto).  Past the end the chunk index is clamped (valid memory, results unused).
    if constexpr (W2ONE) stage_ring(rs_w1, min(j + 1 + D, NCHUNK + 1) * W1_FR * FR, w1s + PH * W1_FR * FR, std::integral_constant<int, W1_FR>{});
    else stage_issue(min(j + 1 + D, NCHUNK + 1), PH, min(j - 1 + D, NCHUNK - 1), (PH + NS - 2) % NS);
    h_init(j + 1, hout);
    const half_t* a1 = w1s + S1 * W1_FR * FR + SW_LANE8;      // W1 chunk j + 1 (for j = NCHUNK - 1: a zero chunk)
    const half_t* a2 = w2s + S2 * W2_FR * FR + SW_LANE8;      // W2 chunk j - 1
    // step i of the sequence: even -> GEMM1 fragment (hb = m & 1, ks = m >> 1), m = i / 2; odd -> GEMM2 fragment ct = i / 2
    auto frag = [&](int i) __attribute__((always_inline)) -> wfrag_t {
      const int m = i >> 1;
      return (i & 1) ? SW_WFRAG(a2 + m * FR) : SW_WFRAG(a1 + ((m & 1) * KS + (m >> 1)) * FR);
    };
#if MQ_SW_SPLIT
    const wfrag_t hs_old = sw_frag(hf_old);                   // the GELU output of chunk j - 1, split once for its CT MFMAs
#else
    const half8& hs_old = hf_old;
#endif
    wfrag_t ring[RD];
    float gts[8], ges[8], gds[8];
#pragma unroll
    for (int i = 0; i < RD && i < N; ++i) ring[i] = frag(i);
    static_for<N>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      static_for<NPIECE>([&](auto qc) __attribute__((always_inline)) {                       // the GELU pieces of this step: piece q belongs to step q N / NPIECE
        constexpr int q = decltype(qc)::value;
        if constexpr (q * N / NPIECE == i)
          gelu_piece(std::integral_constant<int, gelu_piece_of(q, GS, true)>{}, std::integral_constant<int, gelu_piece_of(q, GS, false)>{},
                     hin, hf_new, gts, ges, gds);
      });
      const wfrag_t a = ring[i % RD];
      if constexpr (!(i & 1)) {
        constexpr int m = i >> 1;
        hout[m & 1] = sw_mfma(a, xf[m >> 1], hout[m & 1]);
      } else {
        acc2[i >> 1] = sw_mfma(a, hs_old, acc2[i >> 1]);
      }
      if constexpr (i + RD < N) ring[i % RD] = frag(i + RD);
      __builtin_amdgcn_sched_barrier(0);
    });
    // end of iteration j: the pieces issued at its top must have landed before anybody reads them in iteration j + 1
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * FPW) : "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (W2ONE) {                                    // everybody has read W2 chunk j - 1: chunk j takes its place (exposed: see above)
      stage_ring(rs_w2, min(j, NCHUNK - 1) * W2_FR * FR, w2s, std::integral_constant<int, W2_FR>{});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  };
  float4_ hA[2], hB[2];
  half8 hfA, hfB = zero8();                                   // hfB: H of "chunk -1" = 0 (iteration 0's GEMM2 adds nothing)
  h_init(0, hA);
  gemm1(0, hA);                                               // pipeline fill: H^T of chunk 0
  __syncthreads();                                            // every wave is done with W1 stage 0 before chunk NS lands there
  for (int jb = 0; jb < NCHUNK; jb += U)                      // U iterations: every (stage phase, register set) combination once
    static_for<U>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      if constexpr (u % 2 == 0)
        step(jb + u, std::integral_constant<int, u % NS>{}, std::integral_constant<int, 0>{}, hA, hB, hfB, hfA);
      else
        step(jb + u, std::integral_constant<int, u % NS>{}, std::integral_constant<int, 1>{}, hB, hA, hfA, hfB);
    });
  gemm2(W2ONE ? 0 : (NCHUNK - 1) % NS, hfB);                  // pipeline drain: the last chunk's GEMM 2 (NCHUNK is even: hfB holds it)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the clamped pieces of the last iterations land before the LDS is released

  // ---- epilogue: lane holds OUT^T[c = 16 ct + 4 g + r][token = l15]; + bias + residual (x' re-read: L2-hot), fp32 out; loads of a
  // group of EG channel blocks all in flight before the first add / store (`out` may alias `x`: a lane only re-reads what it writes)
  {
    const long row = row0 + l15;
    const bool live = row < p.M;
    float s = 0.f;
    constexpr int EG = 6;
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
        for (int r = 0; r < 4; ++r) { acc2[ct][r] += (float)b2[i][r] + (xr[i][r] + (float)dl[i][r]); s += acc2[ct][r]; }
      }
#pragma unroll
      for (int i = 0; i < EG; ++i)
        if (live) *(float4_*)(p.out + row * C + (ct0 + i) * 16 + 4 * g) = acc2[ct0 + i];
    }
    if (p.y) {                                                 // fused LayerNorm of the result (next norm1 / stage norm)
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
}

// ---- the TAIL kernel.  swin_mlp2_kernel gives a 16-token block to ONE wave for the whole hidden dimension: a launch lasts
// ceil(blocks / (waves the chip holds)) full passes, and a few blocks over a multiple cost a whole extra pass -- the benchmark's
// C = 384 stage is 8 x 4200 tokens = 2100 blocks on 256 CUs x 8 waves = 2048 slots: 52 blocks too many, two passes, the second one
// on 7 CUs (measured: 0.22 ms; MFMA pipe ~30 % busy within a pass, 14 % over the launch).  Those blocks go to this kernel instead: ONE
// 16-token block per WORKGROUP, the hidden dimension split over its NWT waves (wave w: chunks [w CPW, (w + 1) CPW)), so a block
// takes 1 / NWT of a pass.  Every wave normalises the same 16 tokens, streams ITS chunks' fragment blocks straight from global
// memory (each block is read by exactly one wave of the workgroup: nothing to share, no LDS staging; a register ring keeps RDT
// loads in flight), and holds a partial OUT^T over all C channels; the partials are summed through LDS in a fixed order
// (deterministic), wave w ending up with channel tiles {w, NWT + w, ...}; bias + residual + store + the fused next LayerNorm
// (row statistics across the waves through LDS) follow as in the main kernel.
template <int C, int NWT>
__global__ __launch_bounds__(64 * NWT) void swin_mlp2_tail_kernel(SwinMlp2Params p) {
  constexpr int HID = 4 * C, KS = C / 32, CT = C / 16, NCHUNK = HID / 32, FR = 512;
  constexpr int W1_FR = 2 * KS, W2_FR = CT, N = W1_FR + W2_FR;
  constexpr int CPW = NCHUNK / NWT, TPW = CT / NWT;           // hidden chunks per wave; channel tiles a wave finishes
  // Fragments in flight per wave.  Round 5: the ring only exists if the ISSUE ORDER is pinned -- left alone the scheduler sinks every ring load to
  // just above its use (the ISA of rounds 3-4 waited with vmcnt(1 .. 3): two or three loads in flight, i.e. one HBM / L2 round trip per ~3 KB of
  // a 300 KB stream per wave -- the 64 us of this kernel for a sliver of the tokens); with a scheduling fence behind every refill the waits
  // are vmcnt(RDT).  12 = the deepest ring without a spill at C = 384 (256 VGPRs).
  constexpr int RDT = sizeof(half_t) == 4 ? 6 : 12;
  static_assert(NCHUNK % NWT == 0 && CT % NWT == 0, "tail split");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NH = NWT / 2;                                 // destination waves per half phase
  float4_* red = (float4_*)smem;                              // [NWT src][NH tiles of the half phase][64 lanes]
  float* stat = (float*)(red + NWT * NH * 64);                // [2][NWT][16]: per-wave partial row sums / squared deviations

  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long row0 = (long)blockIdx.x * 16;

#if MQ_SW_SPLIT
  wfrag_t xf[KS];
  {
    half8 xr[KS];
    if (p.delta) ln_fragments<C, true>(p, row0, l15, g, xr); else ln_fragments<C, false>(p, row0, l15, g, xr);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = sw_frag(xr[ks]);       // split once per wave
  }
#else
  half8 xf[KS];
  if (p.delta) ln_fragments<C, true>(p, row0, l15, g, xf); else ln_fragments<C, false>(p, row0, l15, g, xf);
#endif

  float4_ acc2[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) acc2[ct] = (float4_){0.f, 0.f, 0.f, 0.f};

  // fragment t of this wave's stream: chunk wave * CPW + t / N; within a chunk first W1 (hb = m & 1, ks = m >> 1), then W2 (ct)
  const half_t* w1 = p.w1f + (long)wave * CPW * W1_FR * FR + SW_LANE8;
  const half_t* w2 = p.w2f + (long)wave * CPW * W2_FR * FR + SW_LANE8;
  auto frag = [&](int t) __attribute__((always_inline)) -> wfrag_t {
    const int cc = t / N, i = t % N;
    if (i < W1_FR) return SW_WFRAG(w1 + ((long)cc * W1_FR + (i & 1) * KS + (i >> 1)) * FR);
    return SW_WFRAG(w2 + ((long)cc * W2_FR + (i - W1_FR)) * FR);
  };
  wfrag_t ring[RDT];
#pragma unroll
  for (int t = 0; t < RDT; ++t) ring[t] = frag(t);
  static_for<CPW>([&](auto ccc) __attribute__((always_inline)) {
    constexpr int cc = decltype(ccc)::value;
    float4_ h[2];
    {
      const int hb = (wave * CPW + cc) * 32 + 4 * g;
      const half4 b0 = *(const half4*)(p.b1 + hb), b1 = *(const half4*)(p.b1 + hb + 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) { h[0][r] = (float)b0[r]; h[1][r] = (float)b1[r]; }
    }
    static_for<W1_FR>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value, t = cc * N + i;
      h[i & 1] = sw_mfma(ring[t % RDT], xf[i >> 1], h[i & 1]);
      if constexpr (t + RDT < CPW * N) ring[t % RDT] = frag(t + RDT);
      __builtin_amdgcn_sched_barrier(0);
    });
#if MQ_SW_SPLIT
    half8 hf8;
#pragma unroll
    for (int k = 0; k < 8; ++k) hf8[k] = (half_t)gelu_erf2(h[k >> 2][k & 3]);
    const wfrag_t hf = sw_frag(hf8);
#else
    half8 hf;
#pragma unroll
    for (int k = 0; k < 8; ++k) hf[k] = (half_t)gelu_erf2(h[k >> 2][k & 3]);
#endif
    static_for<W2_FR>([&](auto ic) __attribute__((always_inline)) {
      constexpr int ct = decltype(ic)::value, t = cc * N + W1_FR + ct;
      acc2[ct] = sw_mfma(ring[t % RDT], hf, acc2[ct]);
      if constexpr (t + RDT < CPW * N) ring[t % RDT] = frag(t + RDT);
      __builtin_amdgcn_sched_barrier(0);
    });
  });

  // ---- the partials of the NWT waves, summed in wave order; after phase ph wave w holds channel tile ph * NWT + w
  // (two half phases of NWT / 2 destination waves each: the exchange buffer is 32 KB instead of 64 KB, so that a tail workgroup fits on
  // a CU NEXT TO a workgroup of the main kernel -- 107 KB at C = 384 -- when the two run side by side on two streams; same summation order)
  float4_ fin[TPW];
#pragma unroll
  for (int ph = 0; ph < TPW; ++ph) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (ph || hh) __syncthreads();
#pragma unroll
      for (int d = 0; d < NH; ++d) red[(wave * NH + d) * 64 + lane] = acc2[ph * NWT + hh * NH + d];
      __syncthreads();
      if (wave / NH == hh) {                                  // wave-uniform
        const int wl = wave - hh * NH;
        float4_ a = red[wl * 64 + lane];
#pragma unroll
        for (int src = 1; src < NWT; ++src) {
          const float4_ b = red[(src * NH + wl) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r] += b[r];
        }
        fin[ph] = a;
      }
    }
  }

  // ---- epilogue: lane holds OUT^T[c = 16 (ph NWT + wave) + 4 g + r][token l15]
  const long row = row0 + l15;
  const bool live = row < p.M;
  const long rrow = min(row, p.M - 1);
  float s = 0.f;
#pragma unroll
  for (int ph = 0; ph < TPW; ++ph) {
    const int c = (ph * NWT + wave) * 16 + 4 * g;
    const float4_ xr = *(const float4_*)(p.x + rrow * C + c);
    const half4 b2 = *(const half4*)(p.b2 + c);
    half4 dl = (half4){(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    if (p.delta) dl = *(const half4*)(p.delta + rrow * C + c);
#pragma unroll
    for (int r = 0; r < 4; ++r) { fin[ph][r] += (float)b2[r] + (xr[r] + (float)dl[r]); s += fin[ph][r]; }
    if (live) *(float4_*)(p.out + row * C + c) = fin[ph];
  }
  if (p.y) {
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (g == 0) stat[wave * 16 + l15] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NWT; ++w) tot += stat[w * 16 + l15];
    const float mean = tot * (1.f / (float)C);
    float q = 0.f;
#pragma unroll
    for (int ph = 0; ph < TPW; ++ph)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = fin[ph][r] - mean; q += d * d; }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (g == 0) stat[(NWT + wave) * 16 + l15] = q;
    __syncthreads();
    float qt = 0.f;
#pragma unroll
    for (int w = 0; w < NWT; ++w) qt += stat[(NWT + w) * 16 + l15];
    const float rstd = rsqrtf(qt * (1.f / (float)C) + p.eps_n);
    if (live) {
#pragma unroll
      for (int ph = 0; ph < TPW; ++ph) {
        const int c = (ph * NWT + wave) * 16 + 4 * g;
        const half4 gm = *(const half4*)(p.gn + c), bt = *(const half4*)(p.bn + c);
        half4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)((fin[ph][r] - mean) * rstd * (float)gm[r] + (float)bt[r]);
        *(half4*)(p.y + row * C + c) = o;
      }
    }
  }
}

static int device_cus() { return mq_device_cus(); }

template <int C, int NW, bool TABLE>
static int launch_swin_mlp2_main(const SwinMlp2Params& p, hipStream_t s) {
  constexpr bool w2one = MQ_SW_SPLIT && C >= 384;           // (see swin_mlp2_kernel: one W2 stage)
  constexpr size_t smem = (size_t)(2 * 2 * (C / 32) + (w2one ? 1 : 2) * (C / 16)) * 512 * sizeof(half_t) + (size_t)4 * C * sizeof(half_t) +
                          (TABLE ? MQ_GELU_TAB_N * 2 * sizeof(float) : 0);
  static MqOncePerDevice attr;
  if (attr.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)swin_mlp2_kernel<C, NW, TABLE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr.done();
  }
  constexpr int BM = 16 * NW;
  const unsigned grid = (unsigned)((p.M + BM - 1) / BM);
  hipLaunchKernelGGL((swin_mlp2_kernel<C, NW, TABLE>), dim3(grid), dim3(64 * NW), smem, s, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

template <int C, int NWT>
static int launch_swin_mlp2_tail(const SwinMlp2Params& p, hipStream_t s) {
  constexpr size_t smem = (size_t)NWT * (NWT / 2) * 64 * sizeof(float4_) + (size_t)2 * NWT * 16 * sizeof(float);
  static MqOncePerDevice attr;
  if (attr.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)swin_mlp2_tail_kernel<C, NWT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr.done();
  }
  hipLaunchKernelGGL((swin_mlp2_tail_kernel<C, NWT>), dim3((unsigned)((p.M + 15) / 16)), dim3(64 * NWT), smem, s, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

// WPC: workgroups of the main kernel a CU holds (its launch bounds / LDS).  The blocks beyond the last FULL pass go to the tail kernel
// when they are few (at most two rounds of it: 2 x CUs blocks); flags bit 0 switches the split off, bit 2 sends everything to the tail
// kernel (tests).
template <int C, int NW, int NWT, int WPC>
static int dispatch_swin_mlp2(const SwinMlp2Params& p, int flags, hipStream_t s) {
  constexpr int BM = 16 * NW;
  SwinMlp2Params pm = p, pt = p;
  long m_main = p.M;
  if (flags & 4) {
    m_main = 0;
  } else if (!(flags & 1)) {
    const long wgs = (p.M + BM - 1) / BM, slots = (long)device_cus() * WPC, rem = wgs % slots;
    if (wgs > slots && rem > 0 && rem * NW <= 2L * device_cus()) m_main = (wgs - rem) * BM;
  }
  if (m_main < p.M) {
    pt.x = p.x + m_main * C; pt.out = p.out + m_main * C; pt.M = p.M - m_main;
    if (p.delta) pt.delta = p.delta + m_main * C;
    if (p.y) pt.y = p.y + m_main * C;
    pm.M = m_main;
    if (!(flags & 8)) {                                        // bit 3: the caller runs the tail part in a call of its own (other stream)
      const int e = launch_swin_mlp2_tail<C, NWT>(pt, s);
      if (e) return e;
    }
  }
  if (pm.M <= 0 || (flags & 16)) return 0;                     // bit 4: tail part only
  return (flags & 2) ? launch_swin_mlp2_main<C, NW, true>(pm, s) : launch_swin_mlp2_main<C, NW, false>(pm, s);
}

// x [M, C] fp32, delta [M, C] 16-bit or NULL, LN gamma / beta [C], b1 [4C], b2 [C] 16-bit;
// w1f [(4C / 32 + 2) * (C / 16) * 512]: fc1.weight fragment-major -- block (chunk j, hb in {0, 1}, ks) holds for lane l the 8 values
//     W1[32 j + 16 hb + (l & 15)][32 ks + 8 (l >> 4) .. + 7]; two all-zero chunks behind the last one;
// w2f [(4C / 32) * (C / 16) * 512]: fc2.weight with its k-slots permuted (slot 8 g + t of a 32-block <- hidden unit
//     4 g + t for t < 4, 16 + 4 g + t - 4 for t >= 4), fragment-major -- block (chunk j, ct) holds for lane l W2p[16 ct + (l & 15)][32 j + 8 (l >> 4) .. + 7];
// out [M, C] fp32 (may alias x), y [M, C] 16-bit = LayerNorm(out; next_g, next_b, eps_next) if y != NULL.
// flags: bit 1 = table GELU in the main kernel; bit 0 = no tail split; bit 2 = every block through the tail kernel (see dispatch_swin_mlp2);
// bit 3 = only the blocks of the main kernel, bit 4 = only the tail blocks: two calls on two streams run the two parts side by side (they
// touch disjoint rows; a tail workgroup fits beside a main one on a CU).
extern "C" int MQ_SYM(mq_swin_mlp2_fwd)(const float* x, const void* delta, const void* ln_g, const void* ln_b, float eps, const void* w1f,
                                const void* b1, const void* w2f, const void* b2, float* out, const void* next_g, const void* next_b,
                                float eps_next, void* y, long M, int C, int flags, void* stream) {
  if (M <= 0) return 0;
  SwinMlp2Params p;
  p.x = x; p.delta = (const half_t*)delta; p.g2 = (const half_t*)ln_g; p.be2 = (const half_t*)ln_b; p.eps = eps;
  p.w1f = (const half_t*)w1f; p.b1 = (const half_t*)b1; p.w2f = (const half_t*)w2f; p.b2 = (const half_t*)b2; p.out = out;
  p.gn = (const half_t*)next_g; p.bn = (const half_t*)next_b; p.eps_n = eps_next; p.y = (half_t*)y; p.M = M;
  if (y && (!next_g || !next_b)) return -2;
  hipStream_t s = (hipStream_t)stream;
  switch (C) {
#if MQ_SW_SPLIT
    // split-precise: workgroups per CU by the planar rings (48 / 96 / 150 KB of LDS) and the launch bounds; C = 384 with FOUR waves (64 tokens) per
    // workgroup: one wave per SIMD, the planar fragments of 12 k-steps + 24 accumulator tiles need more than 256 registers
    case 96: return dispatch_swin_mlp2<96, 4, 6, 2>(p, flags, s);
    case 192: return dispatch_swin_mlp2<192, 4, 6, 1>(p, flags, s);
    case 384: return dispatch_swin_mlp2<384, 4, 8, 1>(p, flags, s);
#else
    case 96: return dispatch_swin_mlp2<96, 4, 6, 4>(p, flags, s);
    case 192: return dispatch_swin_mlp2<192, 4, 6, 3>(p, flags, s);
    case 384: return dispatch_swin_mlp2<384, 8, 8, 1>(p, flags, s);
#endif
    default: return -1;                                      // other widths: library GEMM path of the caller
  }
}

MQ_NAMESPACE_END
