// mq_gcp_attn_fwd: the whole attention half of a GatedCrossAttentionBlock (modeling_bert_new.py:298-368) in ONE launch -- gfx950, round 5.
//
//   x_out = x + tanh( w2 . gelu( Wg1 LN_g(sup) ) ) * sup ,   sup = Wout . sparse_attn( Wq LN_a(x), kv, idx ) ,   y = LN_f(x_out)
//
// Round 2-4 ran this as LayerNorm, to_q GEMM, mq_gcp_sparse_attn_fwd, to_out GEMM, LayerNorm, gate GEMM, mq_gcp_gate_residual_fwd and the
// LayerNorm in front of the feed-forward half: eight launches of a few microseconds each on the ONE chain of the forward that nothing else
// can overlap (the image-dependent half of the language backbone between the FPN and the first fusion layer), for 34 MFLOP per 16 text rows.
// Everything in that list is ROW-LOCAL once K / V of the vision queries exist (projected once per unique vision token outside, as before):
// a workgroup (8 waves) owns RB = 16 or 32 text rows and walks the chain with its activations in LDS --
//   0  LN_a of the fp32 residual rows -> 16-bit rows in LDS (the rows stay in registers for step 6);
//   1  q = LN_a(x) Wq^T (768 -> 512): wave w owns 64 output columns; ITS weight fragments go global (L2) -> registers in ping-pong groups of
//      k-steps (nobody else reads them: no LDS traffic, no barrier inside the contraction), the A fragments come from the shared rows;
//   2  the sparse gather-attention of mq_gcp_sparse_attn_fwd per row (lane = 8 channels, <= 8 slots, exact-zero rows for tokens without a
//      vision query), in place over q;
//   3  sup = att Wout^T (512 -> 768), rounded to 16 bits like the GEMM output it replaces, into LDS;
//   4  LN_g(sup) -> LDS;   5  h = LN_g(sup) Wg1^T (768 -> 384), gelu, dot with w2 reduced over lanes and waves in a fixed order, tanh;
//   6  x_out = x + gate * sup (fp32 stream), and the LayerNorm the feed-forward half reads next (y, 16-bit).
// Rounding points are those of the unfused chain (every GEMM output / LayerNorm output rounded to the operand type once).
// Weights are streamed per workgroup (2.1 MB from L2, in MFMA B-fragment order -- see gf_rows_gemm): RB = 16 rows for small batches (72
// workgroups at B = 8), RB = 32 for large ones (half the weight traffic per row).
#include "common.h"
#include <type_traits>

MQ_NAMESPACE_BEGIN

namespace {
constexpr int GF_C = 768, GF_HD = 512, GF_G = 384;           // hidden width, heads x dim_head of the cross attention, gate width
constexpr int GF_P = GF_C + 8;                               // LDS row pitch (elements): consecutive rows shift by 16 bytes
constexpr int GF_NT = 512, GF_NW = 8;
constexpr int GF_WKS = 64 * 8;                               // elements between two k-steps of a 16-column tile of the fragment-order weights
}  // namespace

struct GcpAttnParams {
  const float* x;                 // [M, C] fp32 residual stream (M = B * T rows)
  float* x_out;                   // [M, C] (may alias x)
  half_t* y;                      // [M, C] LN_f(x_out) or nullptr
  float* gate_out;                // [M] or nullptr
  const half_t* kv;               // [B, V, 2 * HD]  k | v of the vision queries
  const int* idx;                 // [M, S] indices into V, -1 = padding
  const half_t *wq, *wout, *wg1;  // [HD, C], [C, HD], [G, C] in MFMA B-fragment order: [N / 16][K / 32][64][8]
  const half_t* w2;               // [G]
  const half_t *ga, *ba, *gg, *bg, *gf, *bf;   // LayerNorm gamma / beta: attention input, gate input, feed-forward input
  long M;
  int T, V, S;
  float eps, scale;
};

template <int I, int N, class F>
__device__ __forceinline__ void gf_static_for_impl(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    gf_static_for_impl<I + 1, N>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void gf_static_for(F f) { gf_static_for_impl<0, N>(f); }

// acc[mb][j] += A(rows of this block, K) . W(rows = this wave's 16 j-th output columns, K)^T over NG groups of G k-steps of 32 (K = 32 G NG).
// At: LDS, lane's fragment base (row l15, k offset 8 lg); wrow[j]: global, the lane's 8 elements of the first k-step of its j-th 16-column tile in
// the FRAGMENT-ORDER weights (mqdet_hip.h "MFMA B-fragment order": [N / 16][K / 32][64 lanes][8], k-step stride GF_WKS): one wave instruction
// reads 1 KiB of consecutive bytes.  Reading the same fragments from the row-major [N][K] matrix (16 rows x 64 B per instruction, half a cache
// line per row) streams at 37 GB/s per workgroup against 136 GB/s (tools/probes/l2_weight_stream.hip, round 5 call 16), and the weight stream was
// 55 of this kernel's 73 us at B = 8 (call 15: the launch without it takes 17.5 us, without the MFMAs 71 us).  The weight fragments of a wave are
// read by nobody else: global (L2) -> registers, NSETS register sets, the groups g + 1 .. g + NSETS - 1 requested while group g is multiplied.  The
// loop is unrolled completely and every load group sits behind a scheduling fence: left alone the scheduler sinks the loads to just above their
// first use (the ISA then waits with vmcnt(2): two loads in flight per wave, one L2 / fabric round trip per 2 KB of a 264 KB stream).
template <int NTL, int MB, int G, int NG, int NSETS, int ABL = 0>
__device__ __forceinline__ void gf_rows_gemm(const half_t* At, const half_t* const (&wrow)[NTL], float4_ (&acc)[MB][NTL]) {
  half8 w[NSETS][G][NTL];
  auto load = [&](half8 (&ws)[G][NTL], int grp) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        if constexpr (ABL & 1) ws[g][j] = zero8();                 // ablation: no weight stream
        else ws[g][j] = *(const half8*)(wrow[j] + (grp * G + g) * GF_WKS);
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  gf_static_for<NSETS - 1>([&](auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
    if constexpr (s < NG) load(w[s], s);
  });
  gf_static_for<NG>([&](auto gc) __attribute__((always_inline)) {
    constexpr int grp = decltype(gc)::value;
    if constexpr (grp + NSETS - 1 < NG) load(w[(grp + NSETS - 1) % NSETS], grp + NSETS - 1);
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const half8 af = *(const half8*)(At + mb * 16 * GF_P + (grp * G + g) * 32);
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
          if constexpr (ABL & 8) acc[mb][j][0] += (float)af[0] * (float)w[grp % NSETS][g][j][0];    // ablation: no MFMAs (operands stay live)
          else acc[mb][j] = mfma16(af, w[grp % NSETS][g][j], acc[mb][j]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

// ABL (tools/microbench.py only; results are garbage): the kernel WITHOUT one of its parts -- bit 0: no weight stream (zero fragments), bit 1: no
// gather of the vision keys / values, bit 2: no erf / tanh gate arithmetic, bit 3: no MFMAs -- to see what the 70 us are made of.
template <int MB, int ABL = 0>
__global__ __launch_bounds__(GF_NT, 2) void gcp_attn_kernel(GcpAttnParams p) {
  constexpr int C = GF_C, HD = GF_HD, P = GF_P, RB = 16 * MB, RPW = RB / GF_NW;      // rows per wave in the row-wise steps (2 or 4)
  constexpr int NSETS = MB == 1 ? 3 : 2;         // weight-fragment register sets of the GEMM steps (MB = 2: a third set spills)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* Abuf = (half_t*)smem;                  // [RB][P]: LN_a(x), later sup
  half_t* Bbuf = Abuf + RB * P;                  // [RB][P]: q -> att (first HD columns), later LN_g(sup)
  float* red = (float*)(Bbuf + RB * P);          // [NW][RB] gate partials
  float* gate_s = red + GF_NW * RB;              // [RB]

  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long row0 = (long)blockIdx.x * RB;

  // the vision-query slots of this wave's rows (step 2): requested first, so that their round trip hides under steps 0 / 1
  int idr[RPW][8];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const long tok = min(row0 + wave * RPW + r, p.M - 1);
#pragma unroll
    for (int s = 0; s < 8; ++s) idr[r][s] = s < p.S ? p.idx[tok * p.S + s] : -1;
  }

  // ---- step 0: the block's fp32 rows -> registers (kept for step 6); LN_a -> Abuf.  Wave w owns rows w * RPW .. ; lane holds channels
  // 256 i + 4 lane .. + 3 (i = 0 .. 2): 16-byte loads / stores, a row is three 1 KB pieces.
  float4_ xr[RPW][3];
  auto ln_rows = [&](const float4_ (&v)[RPW][3], const half_t* gam, const half_t* bet, half_t* dst) __attribute__((always_inline)) {
    half4 g4[3], b4[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      g4[i] = *(const half4*)(gam + 256 * i + 4 * lane);
      b4[i] = *(const half4*)(bet + 256 * i + 4 * lane);
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) s += v[r][i][t];
      const float mean = wave_sum(s) * (1.f / C);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) { const float d = v[r][i][t] - mean; q += d * d; }
      const float rstd = rsqrtf(wave_sum(q) * (1.f / C) + p.eps);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        half4 o;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = (half_t)((v[r][i][t] - mean) * rstd * (float)g4[i][t] + (float)b4[i][t]);
        *(half4*)(dst + (wave * RPW + r) * P + 256 * i + 4 * lane) = o;
      }
    }
  };
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const long row = min(row0 + wave * RPW + r, p.M - 1);      // rows beyond M: a copy of the last row (finite; never stored)
#pragma unroll
    for (int i = 0; i < 3; ++i) xr[r][i] = *(const float4_*)(p.x + row * C + 256 * i + 4 * lane);
  }
  ln_rows(xr, p.ga, p.ba, Abuf);
  __syncthreads();

  const half_t* At = Abuf + l15 * P + lg * 8;
  const half_t* Bt = Bbuf + l15 * P + lg * 8;
  // ---- step 1: q = LN_a(x) Wq^T, wave w -> columns 64 w .. 64 w + 63
  {
    float4_ acc[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[mb][j] = (float4_){0.f, 0.f, 0.f, 0.f};
    const half_t* wrow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wrow[j] = p.wq + ((long)(4 * wave + j) * (C / 32) * 64 + lane) * 8;
    gf_rows_gemm<4, MB, 3, C / 32 / 3, NSETS, ABL>(At, wrow, acc);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Bbuf[(mb * 16 + 4 * lg + r) * P + 64 * wave + 16 * j + l15] = (half_t)acc[mb][j][r];
  }
  __syncthreads();

  // ---- step 2: sparse gather-attention, one row at a time per wave (lane = 8 consecutive channels = one eighth of a head), in place
#pragma unroll 2
  for (int r = 0; r < RPW; ++r) {                              // (two rows' gathers in flight together)
    const int lr = wave * RPW + r;
    const long tok = min(row0 + lr, p.M - 1);
    const int b = (int)(tok / p.T);
    half_t* qrow = Bbuf + lr * P + lane * 8;
    const half8 qv = *(const half8*)qrow;
    float qf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = (float)qv[j] * p.scale;
    float sim[8];
    half8 vv[8];
    bool any = false;
    float mx = MQ_NEG_BIG;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int id = (ABL & 2) ? -1 : idr[r][s];
      sim[s] = MQ_NEG_BIG;
      vv[s] = zero8();
      if (id >= 0) {
        const half_t* kr = p.kv + ((long)b * p.V + id) * (2 * HD);
        const half8 kk = *(const half8*)(kr + lane * 8);
        vv[s] = *(const half8*)(kr + HD + lane * 8);
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) d += qf[j] * (float)kk[j];
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        d += __shfl_xor(d, 4);
        sim[s] = d;
        mx = fmaxf(mx, d);
        any = true;
      }
    }
    float acc8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc8[j] = 0.f;
    if (any) {
      float den = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float e = sim[s] > 0.5f * MQ_NEG_BIG ? __expf(sim[s] - mx) : 0.f;
        sim[s] = e;
        den += e;
      }
      const float inv = 1.f / den;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float w = sim[s] * inv;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc8[j] += w * (float)vv[s][j];
      }
    }
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)acc8[j];
    *(half8*)qrow = o;
  }
  __syncthreads();

  // ---- step 3: sup = att Wout^T (K = 512), wave w -> columns 96 w .. 96 w + 95; rounded once, into Abuf (LN_a(x) is dead)
  {
    float4_ acc[MB][6];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[mb][j] = (float4_){0.f, 0.f, 0.f, 0.f};
    const half_t* wrow[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) wrow[j] = p.wout + ((long)(6 * wave + j) * (HD / 32) * 64 + lane) * 8;
    gf_rows_gemm<6, MB, 2, HD / 32 / 2, NSETS, ABL>(Bt, wrow, acc);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Abuf[(mb * 16 + 4 * lg + r) * P + 96 * wave + 16 * j + l15] = (half_t)acc[mb][j][r];
  }
  __syncthreads();

  // ---- step 4: LN_g(sup) -> Bbuf (att is dead: every wave passed the barrier behind step 3)
  float4_ sr[RPW][3];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const half4 h4 = *(const half4*)(Abuf + (wave * RPW + r) * P + 256 * i + 4 * lane);
#pragma unroll
      for (int t = 0; t < 4; ++t) sr[r][i][t] = (float)h4[t];
    }
  ln_rows(sr, p.gg, p.bg, Bbuf);
  __syncthreads();

  // ---- step 5: h = LN_g(sup) Wg1^T (N = 384), wave w -> columns 48 w .. 48 w + 47; gate = tanh(sum_c gelu(round16(h_c)) w2_c)
  {
    float4_ acc[MB][3];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[mb][j] = (float4_){0.f, 0.f, 0.f, 0.f};
    const half_t* wrow[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) wrow[j] = p.wg1 + ((long)(3 * wave + j) * (C / 32) * 64 + lane) * 8;
    gf_rows_gemm<3, MB, 4, C / 32 / 4, NSETS, ABL>(Bt, wrow, acc);
    float w2v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) w2v[j] = (float)p.w2[48 * wave + 16 * j + l15];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float v = (float)(half_t)acc[mb][j][r];                 // the rounding point of the gate GEMM's output
          if constexpr (ABL & 4) d += v * w2v[j];
          else d += 0.5f * v * (1.f + erff(v * 0.70710678118654752f)) * w2v[j];
        }
        d = group16_sum(d);
        if (l15 == 0) red[wave * RB + mb * 16 + 4 * lg + r] = d;
      }
  }
  __syncthreads();
  if (tid < RB) {
    float d = 0.f;
#pragma unroll
    for (int w = 0; w < GF_NW; ++w) d += red[w * RB + tid];
    const float gate = (ABL & 4) ? d : tanhf(d);
    gate_s[tid] = gate;
    if (p.gate_out && row0 + tid < p.M) p.gate_out[row0 + tid] = gate;
  }
  __syncthreads();

  // ---- step 6: x_out = x + gate * sup (fp32 stream), y = LN_f(x_out)
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const float gate = gate_s[wave * RPW + r];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) xr[r][i][t] = sr[r][i][t] * gate + xr[r][i][t];
    const long row = row0 + wave * RPW + r;
    if (row < p.M) {
#pragma unroll
      for (int i = 0; i < 3; ++i) *(float4_*)(p.x_out + row * C + 256 * i + 4 * lane) = xr[r][i];
    }
  }
  if (p.y) {
    // LN_f through the same routine into Bbuf (LN_g(sup) is dead behind the barrier above), then whole rows out
    __syncthreads();
    ln_rows(xr, p.gf, p.bf, Bbuf);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const long row = row0 + wave * RPW + r;
      if (row < p.M) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
          *(half4*)(p.y + row * C + 256 * i + 4 * lane) = *(const half4*)(Bbuf + (wave * RPW + r) * P + 256 * i + 4 * lane);
      }
    }
  }
}

template <int MB, int ABL = 0>
static int launch_gcp_attn(const GcpAttnParams& p, hipStream_t stream) {
  constexpr int RB = 16 * MB;
  constexpr size_t smem = (size_t)2 * RB * GF_P * sizeof(half_t) + (size_t)(GF_NW + 1) * RB * sizeof(float);
  static MqOncePerDevice attr;
  if (attr.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)gcp_attn_kernel<MB, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr.done();
  }
  hipLaunchKernelGGL((gcp_attn_kernel<MB, ABL>), dim3((unsigned)((p.M + RB - 1) / RB)), dim3(GF_NT), smem, stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

// x [B*T, 768] fp32 (the text residual stream), x_out the same shape (may alias x), y [B*T, 768] 16-bit = LN_f(x_out) or NULL, gate_out [B*T] or
// NULL; kv [B, V, 1024] 16-bit = to_kv(norm_kv(vision)) (k | v); idx [B*T, S] int32 (S <= 8; -1 = padding); wq [512, 768], wout [768, 512],
// wg1 [384, 768], w2 [384]; ln = {gamma, beta} x {attention input, gate input, feed-forward input}, 768 each.  rows_per_block: 16 or 32 (0: chosen
// from M).  Returns -1 for other widths / S > 8.
extern "C" int MQ_SYM(mq_gcp_attn_fwd)(const float* x, float* x_out, void* y, float* gate_out, const void* kv, const int* idx, const void* wq,
                                       const void* wout, const void* wg1, const void* w2, const void* ln_a_g, const void* ln_a_b,
                                       const void* ln_g_g, const void* ln_g_b, const void* ln_f_g, const void* ln_f_b, long M, int T, int V,
                                       int S, int C, int heads, int dim_head, int G, float eps, int rows_per_block, void* stream) {
  if (M <= 0) return 0;
  if (C != GF_C || heads * dim_head != GF_HD || dim_head != 64 || G != GF_G || S < 0 || S > 8 || T <= 0) return -1;
  GcpAttnParams p;
  p.x = x; p.x_out = x_out; p.y = (half_t*)y; p.gate_out = gate_out; p.kv = (const half_t*)kv; p.idx = idx;
  p.wq = (const half_t*)wq; p.wout = (const half_t*)wout; p.wg1 = (const half_t*)wg1; p.w2 = (const half_t*)w2;
  p.ga = (const half_t*)ln_a_g; p.ba = (const half_t*)ln_a_b; p.gg = (const half_t*)ln_g_g; p.bg = (const half_t*)ln_g_b;
  p.gf = (const half_t*)ln_f_g; p.bf = (const half_t*)ln_f_b;
  p.M = M; p.T = T; p.V = V; p.S = S; p.eps = eps; p.scale = 1.0f / sqrtf((float)dim_head);
#ifdef MQ_PRIMARY_UNIT
  switch (rows_per_block >> 8) {      // diagnostic only (tools/microbench.py): 16 | mask << 8 launches the kernel WITHOUT the parts in the mask; the output is meaningless
    case 1: return launch_gcp_attn<1, 1>(p, (hipStream_t)stream);
    case 2: return launch_gcp_attn<1, 2>(p, (hipStream_t)stream);
    case 4: return launch_gcp_attn<1, 4>(p, (hipStream_t)stream);
    case 8: return launch_gcp_attn<1, 8>(p, (hipStream_t)stream);
    case 15: return launch_gcp_attn<1, 15>(p, (hipStream_t)stream);
    default: break;
  }
#endif
  if (rows_per_block == 0) rows_per_block = (M >= 32L * 2 * mq_device_cus() && sizeof(half_t) == 2) ? 32 : 16;
  if (rows_per_block == 32 && sizeof(half_t) == 2) return launch_gcp_attn<2>(p, (hipStream_t)stream);
  return launch_gcp_attn<1>(p, (hipStream_t)stream);
}

MQ_NAMESPACE_END
