// VLFuse bi-directional cross attention (BiMultiHeadAttention, reference utils/fuse_helper.py:218-303) for gfx950:
// two kernels specialised for 8 heads x 256, image tokens N ~ 22400 / image, text tokens T <= 256, with the image-side
// projections folded into the text operands, so that both read LN(v) [B, N, 256] directly.
//
//   mq_vlfuse_i2t_fwd   (image side):  out[b,n,:] = LN(v)[b,n,:] + ob + sum_h softmax_t( clamp(LN(v)[b,n].Kf[b,h,t] + bias[b,h,t]) ) Vo[b,h,t,:]
//   mq_vlfuse_t2i_fwd   (text side):   out[b,t,h,:] = softmax_n( clamp(Kf[b,h,t].LN(v)[b,n]) ) LN(v)[b,n,:]
//
// What the generic mq_attn_fwd (attn.hip) left on the table for these two shapes (profiles/README.md):
//   * image side: one workgroup per (q-tile, head) re-read the 64 KB Q tile and wrote a 64 KB per-head output for each
//     of the 8 heads (1.5 GB of HBM traffic per launch, then a separate head-sum kernel read it all back).  Here a
//     workgroup keeps its Q fragments in registers, loops over the 8 heads, accumulates sum_h P_h Vo_h in ONE fp32
//     accumulator (exact softmax per head: all <= 256 logits of a row stay in registers) and writes the final
//     residual-added [128, 256] tile once: 92 MB read + 92 MB written per launch.
//   * text side: keys and values are the SAME rows (LN(v)); the generic kernel loaded them twice (row-major K tile and
//     a strided V^T tile).  Here one row-major tile in LDS feeds both MFMAs: QK^T reads it with ds_read_b128, PV reads
//     it TRANSPOSED with ds_read_b64_tr_b16.
//   * both: S^T = K Q^T is computed instead of S (operands swapped), so a lane that owns column `query` of S^T already
//     holds exactly the P^T B-fragment the PV MFMA needs (k-slot (lg, j) <-> key 16*(j/4) + 4*lg + j%4, the same
//     permutation the transposed V read produces): P never goes through LDS, row reductions are 2 shuffles.
//   * both: XCD-aware work order (all workgroups streaming the same tiles sit on one XCD -> its L2 serves them),
//     two-slot register prefetch ring (tiles u+1 and u+2 in flight while tile u is on the MFMAs), double-buffered LDS
//     tiles with row pitch 272 halfs (conflict-free for both the b128 and the tr_b16 reads), one barrier per tile.
#include "common.h"
#include <type_traits>
#include <cstdlib>

MQ_NAMESPACE_BEGIN

// SPLIT-PRECISE build (-DMQ_F32, round 6).  Round 5 compiled these kernels with fp32 tiles and split every fragment inside mfma16: 450 ... 1200 spilled
// VGPRs, 8.0 / 6.3 ms per launch against 0.34 / 0.26 ms with fp16 operands.  Now the LDS tiles are PLANAR -- a tile of 64 rows x 256 operand
// elements is two fp16 planes [64][KS], hi = fp16(x) and lo = fp16((x - hi) 2^11) (csrc/common.h), the same bytes as the fp32 tile -- written by
// the thread that stages a chunk (one split per element and workgroup), so a fragment read is one ds_read_b128 per plane, the transposed value
// reads are the native ds_read_b64_tr_b16 on each plane, the Q fragments are split once when they are loaded, the P fragments once per 32-key
// step, and every contraction is three v_mfma_f32_16x16x32_f16 (mfma16_split).  One ring slot instead of two (the fp32 chunks of a tile are twice
// the registers).
#if defined(MQ_F32) && !defined(MQ_F32_EXACT)
#define MQ_VL_SPLIT 1
typedef mq_split8 vfrag;                   // an MFMA operand fragment: (hi, lo) fp16 x 8
__device__ __forceinline__ vfrag vl_frag(half8 x) { return mq_split(x); }
__device__ __forceinline__ float4_ vl_mfma(const vfrag& a, const vfrag& b, float4_ c) { return mfma16_split(a, b, c); }
#else
#define MQ_VL_SPLIT 0
typedef half8 vfrag;
__device__ __forceinline__ vfrag vl_frag(half8 x) { return x; }
__device__ __forceinline__ float4_ vl_mfma(vfrag a, vfrag b, float4_ c) { return mfma16(a, b, c); }
#endif

namespace {
constexpr int VH = 8, VD = 256;            // max heads (run-time count in the params), head dim
constexpr int BM = 128;                    // query rows per workgroup (4 waves x 32)
constexpr int TK = 64;                     // keys per tile
constexpr int KS = VD + 16;                // LDS row pitch (halfs): 544 B -> 8 consecutive rows cover all 64 banks
constexpr int TILE = TK * KS;              // halfs per LDS tile
using S0 = std::integral_constant<int, 0>;
using S1 = std::integral_constant<int, 1>;
}  // namespace

struct I2TParams {
  const half_t* v;        // [B, N, 256] LN(v): queries and residual
  const half_t* kf;       // [B, 8, T, 256] folded text keys    } element (b, h, t, :) at b * kv_bs + h * kv_hs + t * kv_ts (elements): a contiguous
  const half_t* vo;       // [B, 8, T, 256] folded text values  } [B, 8, T, 256] tensor or a view of the projection GEMM's [B, T, 8 x 256 | ...] output
  long kv_bs, kv_hs, kv_ts;
  const float* bias;      // [B, 8, T] additive logit bias; <= -1e29 marks a masked key
  const int* kv_len;      // [B] or nullptr: keys >= kv_len[b] are masked
  const half_t* obias;    // [256]
  half_t* out;            // [B, N, 256]
  int B, N, T;
  int H;                  // heads (8: VLDyHead fusion, 4: GroundingDINO feature enhancer), <= VH
  float clamp;
};

// One tile = 64 rows x 256 halfs (32 KB): 2048 16-byte chunks, 8 per thread with 256 threads, 4 with 512.
// QB = 16-query blocks per wave: 2 -> 4 waves x 32 rows (one wave per SIMD), 1 -> 8 waves x 16 rows (two per SIMD: a second
// wave hides the LDS latency and the softmax VALU work of the first; each wave then re-reads the K / V fragments for
// half as many MFMAs, which the LDS still sustains).
template <int NTH> struct TileRegs { half8 r[2048 / NTH]; };

template <int I, int N, class F>
__device__ __forceinline__ void static_for_u_impl(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_u_impl<I + 1, N>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void static_for_u(F f) { static_for_u_impl<0, N>(f); }

// NCH: chunk rounds to do -- round i covers tile rows [i * NTH / 32, (i + 1) * NTH / 32): a tile whose tail rows nobody reads
// (the last key / value tile of a caption that ends inside it) is staged only as far as it is read
template <int NTH, int NCH = 2048 / NTH>
__device__ __forceinline__ void tile_issue(TileRegs<NTH>& t, const half_t* src, int row0, int last_row, int tid, int ld = VD) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NTH;
    const int row = min(row0 + (c >> 5), last_row);
    t.r[i] = *(const half8*)(src + (row * ld + (c & 31) * 8));           // ld: row pitch of the source in elements (a tile's rows: < 2^31)
  }
}
template <int NTH, int NCH = 2048 / NTH>
__device__ __forceinline__ void tile_commit(const TileRegs<NTH>& t, half_t* dst, int tid) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NTH;
#if MQ_VL_SPLIT
    const mq_split8 sp = mq_split(t.r[i]);                 // 8 fp32 operands -> 16 bytes into the hi plane, 16 into the lo plane
    _Float16* h = (_Float16*)dst + (c >> 5) * KS + (c & 31) * 8;
    *(mq_h16x8*)h = sp.hi;
    *(mq_h16x8*)(h + TILE) = sp.lo;
#else
    *(half8*)(dst + (c >> 5) * KS + (c & 31) * 8) = t.r[i];
#endif
  }
}

// S^T[nb][qb] = K_tile . Q^T : A = K rows (keys) from LDS, B = Q fragments from registers (QLDS: from the LDS copy)
// NBL: 16-key blocks of this tile that hold at least one un-masked key (compile time: the caller branches, wave-uniformly,
// on the block count of the LAST tile only); the others keep s = 0 and are masked by the caller -- a 141-token caption
// fills 9 of the 12 blocks of its three 64-key tiles
template <bool QLDS, int QB, int NBL = 4>
__device__ __forceinline__ void qk_tile(const half_t* tile, const vfrag (&qf)[QB][8], const half_t* qw, float4_ (&s)[4][QB],
                                        int l15, int lg) {
  static_assert(!(MQ_VL_SPLIT && QLDS), "split-precise: the Q fragments live in registers (a planar Q tile does not fit beside the key tiles)");
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) s[nb][qb] = (float4_){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < VD / 32; ++kk) {
    vfrag kf[4], qq[QB];
#pragma unroll
    for (int nb = 0; nb < NBL; ++nb) {
#if MQ_VL_SPLIT
      const _Float16* kp = (const _Float16*)tile + (nb * 16 + l15) * KS + kk * 32 + lg * 8;
      kf[nb].hi = *(const mq_h16x8*)kp;
      kf[nb].lo = *(const mq_h16x8*)(kp + TILE);
#else
      kf[nb] = *(const half8*)(tile + (nb * 16 + l15) * KS + kk * 32 + lg * 8);
#endif
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#if !MQ_VL_SPLIT
      if constexpr (QLDS) qq[qb] = *(const half8*)(qw + (qb * 16 + l15) * KS + kk * 32 + lg * 8);
      else
#endif
      qq[qb] = qf[qb][kk];
    }
#pragma unroll
    for (int nb = 0; nb < NBL; ++nb)
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) s[nb][qb] = vl_mfma(kf[nb], qq[qb], s[nb][qb]);
  }
}

// O^T[db][qb] += V_tile^T . P^T : A = transposed reads of the row-major tile, B = P^T fragments from registers
// STL: 32-key steps of this tile with a non-zero probability (compile time, see qk_tile)
template <int QB, int STL = 2>
__device__ __forceinline__ void pv_tile(const half_t* tile, const half8 (&pf)[2][QB], float4_ (&o)[16][QB], int l15, int lg) {
#pragma unroll
  for (int st = 0; st < STL; ++st) {
#if MQ_VL_SPLIT
    const _Float16* base = (const _Float16*)tile + (st * 32 + 4 * lg + (l15 >> 2)) * KS + (l15 & 3) * 4;
    vfrag pq[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) pq[qb] = mq_split(pf[st][qb]);      // once per 32-key step: feeds the 16 channel blocks below
#pragma unroll
    for (int db = 0; db < 16; ++db) {
      vfrag a;
      const mq_h16x4_t h0 = lds_read_tr16_h(base + db * 16), h1 = lds_read_tr16_h(base + 16 * KS + db * 16);
      const mq_h16x4_t l0 = lds_read_tr16_h(base + TILE + db * 16), l1 = lds_read_tr16_h(base + TILE + 16 * KS + db * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) { a.hi[j] = h0[j]; a.hi[4 + j] = h1[j]; a.lo[j] = l0[j]; a.lo[4 + j] = l1[j]; }
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) o[db][qb] = vl_mfma(a, pq[qb], o[db][qb]);
    }
#else
    const half_t* base = tile + (st * 32 + 4 * lg + (l15 >> 2)) * KS + (l15 & 3) * 4;
#pragma unroll
    for (int db = 0; db < 16; ++db) {
      const half4 lo = lds_read_tr16(base + db * 16);
      const half4 hi = lds_read_tr16(base + 16 * KS + db * 16);
      half8 a;
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = lo[j]; a[4 + j] = hi[j]; }
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) o[db][qb] = mfma16(a, pf[st][qb], o[db][qb]);
    }
#endif
  }
}

// NT = number of 64-key tiles whose logits stay in registers.  NT <= 2 (<= 128 text tokens): Q fragments in registers and a
// two-slot prefetch ring.  NT >= 3 ("lean"): the logits alone take 96-128 VGPRs, so Q moves to LDS and the ring has one slot.
// NBL = live 16-key blocks of the LAST tile (1..4), chosen by the host from max_kv: a 141-token caption is NT = 3, NBL = 1
// (144 keys instead of 192).  Compile time on purpose: any run-time branch around the MFMA loops pushes this kernel, which
// sits at the 256-VGPR limit, into spills.  Batch items with shorter captions are handled by the key mask as before.
// ABL (tools/microbench.py "vlfuse" only; results are garbage): time the kernel WITHOUT one of its parts -- bit 0: no global tile loads,
// bit 1: no LDS tile commits, bit 2: no softmax arithmetic, bit 3: no fragment reads / MFMAs -- to see which part a step waits for.
// QREG (NT = 3 with at most two live blocks in the last tile, i.e. 129 .. 160 keys -- the 141-token caption of the benchmark): the Q
// fragments stay in registers and the prefetch ring keeps its two slots as for NT <= 2 (the 36 .. 40 live logit registers leave the
// room: 242 / 246 VGPRs, no spill) -- no Q tile in LDS, a fifth fewer fragment reads in the QK steps, tiles two steps ahead.
template <int NT, int QB, int NBL, int ABL = 0, bool QREG = false>
__global__ __launch_bounds__(2048 / (QB * 4)) void vlfuse_i2t_kernel(I2TParams p) {
  constexpr bool LEAN = NT >= 3 && !QREG && !MQ_VL_SPLIT;   // (split-precise: Q always in registers)
  constexpr bool ONE = LEAN || MQ_VL_SPLIT;                // one ring slot: the next tile only
  constexpr int NTH = 2048 / (QB * 4), WR = 16 * QB;        // threads per workgroup (512 / 256), query rows per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* tiles = (half_t*)smem;                           // [2][TK][KS]   (split-precise: [2][hi | lo][TK][KS] fp16 -- the same bytes)
  half_t* Qs = tiles + 2 * TILE;                           // [BM][KS] (LEAN only)
  float* bias_s = (float*)(Qs + (LEAN ? BM * KS : 0));     // [8][NT*64], + the additive key mask [8][NT*64] behind it

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  // XCD-aware order: workgroup i runs on XCD i % 8.  The (image, q-tile) list is cut into 8 CONTIGUOUS chunks, one per XCD: with B = 8
  // XCD x takes exactly image x (its text operands stay in that L2), with B = 4 two XCDs share an image, with B = 16 an XCD takes two --
  // every XCD is busy for any B (round 3 mapped image b to XCD b % 8: at B = 4, the batch of BASELINE configs[3], half of the chip idled)
  const int qtiles = (p.N + BM - 1) / BM;
  const long total = (long)p.B * qtiles;
  const int per = (int)((total + 7) >> 3);
  const int wslot = blockIdx.x >> 3;
  const long w = (long)(blockIdx.x & 7) * per + wslot;
  if (wslot >= per || w >= total) return;
  const int b = (int)(w / qtiles);
  const int qtile = (int)(w - (long)b * qtiles);
  const int row0 = qtile * BM + wave * WR;

  // bias_s = bias, 0 for a masked key; mask_s = 0 or -1e30 -- added AFTER the clamp, like the reference's attention mask behind torch.clamp
  // (fuse_helper.py:236-262): logit = med3(s + bias, -c, c) + mask, three VALU operations (round 3: add, min, max, compare, two selects).
  // (Not in the log2 domain: with the raw v_exp_f32 builtin hipcc 7.0 spills 237 VGPRs in this kernel; __expf costs one multiply more.)
  float* mask_s = bias_s + VH * NT * TK;
  const int kv_eff = p.kv_len ? max(1, min(p.T, p.kv_len[b])) : p.T;
  for (int i = tid; i < p.H * NT * TK; i += NTH) {
    const int h = i / (NT * TK), t = i % (NT * TK);
    float v = MQ_NEG_BIG;
    if (t < kv_eff) v = p.bias ? p.bias[((long)b * p.H + h) * p.T + t] : 0.f;
    const bool masked = v < -1.0e29f;
    bias_s[i] = masked ? 0.f : v;
    mask_s[i] = masked ? MQ_NEG_BIG : 0.f;
  }
  const float cl2 = p.clamp > 0.f ? p.clamp : 3.0e38f;                   // no clamp: a bound no logit reaches (one v_med3 either way)

  vfrag qf[QB][8];
  const half_t* vb = p.v + (long)b * p.N * VD;
  const half_t* qw = Qs + wave * WR * KS;
  if constexpr (LEAN) {
    TileRegs<NTH> q0, q1;                                        // 128 rows = two 64-row tiles
    tile_issue(q0, vb, qtile * BM, p.N - 1, tid);
    tile_issue(q1, vb, qtile * BM + 64, p.N - 1, tid);
    tile_commit(q0, Qs, tid);
    tile_commit(q1, Qs + 64 * KS, tid);
  } else {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int row = min(row0 + qb * 16 + l15, p.N - 1);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) qf[qb][kk] = vl_frag(*(const half8*)(vb + (long)row * VD + kk * 32 + lg * 8));
    }
  }

  float4_ o[16][QB];
#pragma unroll
  for (int db = 0; db < 16; ++db)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) o[db][qb] = (float4_){0.f, 0.f, 0.f, 0.f};

  // tile stream: u = h * 2NT + j;  j < NT: key tile j of head h,  j >= NT: value tile j - NT.  2NT is even, so the
  // position parity inside a head is also the parity of u: LDS buffer and register slot indices are compile-time.
  constexpr int PER_HEAD = 2 * NT;
  const int U = p.H * PER_HEAD;
  TileRegs<NTH> slot[ONE ? 1 : 2];
  if constexpr (ABL != 0) {
#pragma unroll
    for (int i = 0; i < 2048 / NTH; ++i) { slot[0].r[i] = zero8(); slot[ONE ? 0 : 1].r[i] = zero8(); }
  }
  // chunk rounds of the tile at stream position np (0 .. 2 NT - 1; K tiles first): the last K tile is read up to its NBL live 16-key
  // blocks, the last V tile up to the 32-key steps that hold one (rows beyond stay whatever the buffer held before: never read)
  auto rounds_of = [](int np) constexpr {                  // np < 0: a whole tile (pipeline fill)
    if (np < 0) return 2048 / NTH;
    np %= 2 * NT;
    const int j = np % NT, rows = j != NT - 1 ? 64 : (np < NT ? 16 * NBL : 32 * ((NBL + 1) / 2));
    return (rows * 32 + NTH - 1) / NTH;
  };
  auto issue = [&](auto SLOT, auto NP, int u) {
    constexpr int sl = decltype(SLOT)::value, np = decltype(NP)::value;
    u = min(u, U - 1);                                     // the tail re-loads the last tile: one code path, no branches
    const int h = u / PER_HEAD, j = u % PER_HEAD;
    const half_t* src = (j < NT ? p.kf : p.vo) + (long)b * p.kv_bs + (long)h * p.kv_hs;
    if constexpr (!(ABL & 1)) tile_issue<NTH, rounds_of(np)>(slot[sl], src, (j < NT ? j : j - NT) * TK, p.T - 1, tid, (int)p.kv_ts);
  };
  // begin(pos): prefetch;  end(pos): commit the next tile into the other LDS buffer + barrier
  auto begin = [&](auto POS, int u) {
    constexpr int pos = decltype(POS)::value, par = pos & 1;
    if constexpr (ONE) issue(S0{}, std::integral_constant<int, pos + 1>{}, u + 1);
    else issue(std::integral_constant<int, par>{}, std::integral_constant<int, pos + 2>{}, u + 2);  // the slot of tile u was committed one step ago
    __builtin_amdgcn_sched_barrier(0);                     // keep the prefetch ahead of everything that waits on VMEM
  };
  auto end = [&](auto POS) {
    constexpr int pos = decltype(POS)::value, par = pos & 1;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 2)) tile_commit<NTH, rounds_of((pos + 1) % PER_HEAD)>(slot[ONE ? 0 : (par ^ 1)], tiles + (par ^ 1) * TILE, tid);
    __syncthreads();
  };
  issue(S0{}, std::integral_constant<int, -1>{}, 0);       // the fill stages whole tiles
  if constexpr (!ONE) issue(S1{}, std::integral_constant<int, -1>{}, 1);
  tile_commit(slot[0], tiles, tid);
  __syncthreads();

  for (int h = 0; h < p.H; ++h) {
    const int u0 = h * PER_HEAD;
    float4_ s[NT][4][QB];
    // ---- logits of all key tiles of this head (kept in registers: exact softmax, no running rescale)
    auto qk_step = [&](auto J) {
      constexpr int j = decltype(J)::value;
      begin(J, u0 + j);
      if constexpr (!(ABL & 8)) qk_tile<LEAN, QB, (j == NT - 1 ? NBL : 4)>(tiles + (j & 1) * TILE, qf, qw, s[j], l15, lg);
      else {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) s[j][nb][qb] = (float4_){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int nb = 0; nb < (j == NT - 1 ? NBL : 4); ++nb) {           // live key blocks only (the others never reach the softmax)
        const float4_ kb = *(const float4_*)(bias_s + (h * NT + j) * TK + nb * 16 + 4 * lg);
        const float4_ km = *(const float4_*)(mask_s + (h * NT + j) * TK + nb * 16 + 4 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
            s[j][nb][qb][r] = __builtin_amdgcn_fmed3f(s[j][nb][qb][r] + kb[r], -cl2, cl2) + km[r];
      }
      end(J);
    };
    qk_step(std::integral_constant<int, 0>{});
    if constexpr (NT > 1) qk_step(std::integral_constant<int, 1>{});
    if constexpr (NT > 2) qk_step(std::integral_constant<int, 2>{});
    if constexpr (NT > 3) qk_step(std::integral_constant<int, 3>{});

    // ---- softmax over the text keys: a lane owns query column l15 of block qb; its keys are spread over (j, nb, r)
    // in-lane and over the 4 lane groups lg.  s <- exp(s - max) / sum
#pragma unroll
    for (int qb = 0; qb < ((ABL & 4) ? 0 : QB); ++qb) {
      float mx = MQ_NEG_BIG;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int nb = 0; nb < (j == NT - 1 ? NBL : 4); ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[j][nb][qb][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int nb = 0; nb < (j == NT - 1 ? NBL : 4); ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __expf(s[j][nb][qb][r] - mx);
            s[j][nb][qb][r] = e;
            sum += e;
          }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      const float inv = __builtin_amdgcn_rcpf(sum);                       // sum >= 1 (the row maximum contributes 1): no denormal / zero case
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int nb = 0; nb < (j == NT - 1 ? NBL : 4); ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[j][nb][qb][r] *= inv;
    }
    // ---- O^T += Vo_h^T P_h^T over the value tiles of this head (P^T fragments are built from s[j] right before use)
    auto pv_step = [&](auto J) {
      constexpr int j = decltype(J)::value;
      using POS = std::integral_constant<int, NT + j>;
      begin(POS{}, u0 + NT + j);
      half8 pf[2][QB];
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            constexpr int live = (j == NT - 1 ? NBL : 4);
            pf[st][qb][r] = 2 * st < live ? (half_t)s[j][2 * st][qb][r] : (half_t)0.f;
            pf[st][qb][4 + r] = 2 * st + 1 < live ? (half_t)s[j][2 * st + 1][qb][r] : (half_t)0.f;
          }
      if constexpr (!(ABL & 8)) pv_tile<QB, (j == NT - 1 ? (NBL + 1) / 2 : 2)>(tiles + ((NT + j) & 1) * TILE, pf, o, l15, lg);
      else {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[0][qb][r] += (float)pf[0][qb][r] + (float)pf[1][qb][4 + r];
      }
      end(POS{});
    };
    pv_step(std::integral_constant<int, 0>{});
    if constexpr (NT > 1) pv_step(std::integral_constant<int, 1>{});
    if constexpr (NT > 2) pv_step(std::integral_constant<int, 2>{});
    if constexpr (NT > 3) pv_step(std::integral_constant<int, 3>{});
  }

  // ---- epilogue: O^T -> LDS (row = query), + residual LN(v) + out-proj bias, 16-byte coalesced stores
  constexpr int OS = VD + 8;
  half_t* Os = tiles + wave * (WR * OS);                   // [waves][WR][OS] aliases the tiles (all waves passed the last barrier)
#pragma unroll
  for (int db = 0; db < 16; ++db)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      half4 v4;
#pragma unroll
      for (int r = 0; r < 4; ++r) v4[r] = (half_t)o[db][qb][r];
      *(half4*)(Os + (qb * 16 + l15) * OS + db * 16 + 4 * lg) = v4;
    }
  wave_lds_fence();
  half_t* ob = p.out + (long)b * p.N * VD;
#pragma unroll
  for (int i = 0; i < 8 * QB; ++i) {
    const int c = lane + i * 64;
    const int rr = c >> 5, ch = c & 31;
    const int row = row0 + rr;
    if (row < p.N) {
      const half8 a = *(const half8*)(Os + rr * OS + ch * 8);
      const half8 res = *(const half8*)(vb + (long)row * VD + ch * 8);
      const half8 bb = *(const half8*)(p.obias + ch * 8);
      half8 y;
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = (half_t)((float)a[j] + (float)res[j] + (float)bb[j]);
      *(half8*)(ob + (long)row * VD + ch * 8) = y;
    }
  }
}

template <int NT, int QB, int NBL, int ABL = 0, bool QREG = false>
static int launch_i2t(const I2TParams& p, hipStream_t stream) {
  constexpr size_t smem = (size_t)(2 * TILE + ((NT >= 3 && !QREG && !MQ_VL_SPLIT) ? BM * KS : 0)) * sizeof(half_t) + (size_t)2 * VH * NT * TK * sizeof(float);
  static_assert(4 * 32 * (VD + 8) <= 2 * TILE, "O staging must fit in the tiles");
  static MqOncePerDevice attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)vlfuse_i2t_kernel<NT, QB, NBL, ABL, QREG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set.done();
  }
  const int qtiles = (p.N + BM - 1) / BM;
  const long per = ((long)p.B * qtiles + 7) / 8;
  hipLaunchKernelGGL((vlfuse_i2t_kernel<NT, QB, NBL, ABL, QREG>), dim3((unsigned)(8 * per)), dim3(2048 / (QB * 4)), smem, stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

// (Round 3 built a PAIR-SPLIT variant of the image side: two waves share 32 query rows, each computes the logits of every second
// 16-key block and the outputs of half of the 256 channels -- Q fragments in registers, half the LDS fragment reads per MFMA,
// softmax statistics and P exchanged through LDS.  Measured (profiles/r03_call{5,6,8}_microbench_vlfuse*.json): 1.5x slower with three
// key tiles (Q + logits + outputs exceed 256 VGPRs: 92 spills), +4.6 % with two -- and no better than this kernel once its last tile
// was staged only as far as it is read.  Halving the LDS reads alone does not pay here; removed.)

// waves x rows shape of the text-side kernel: 8 waves x 16 query rows (default) or 4 waves x 32 (MQ_VLFUSE_QB=2, A/B switch)
static int vlfuse_qb() {
  static const int qb = [] { const char* e = getenv("MQ_VLFUSE_QB"); return (e && e[0] == '2') ? 2 : 1; }();
  return qb;
}

// Image side of VLFuse.  max_kv: host-known upper bound of kv_len (T if unknown) -- picks the number of 64-key tiles
// kept in registers.  See include/mqdet_hip.h.
extern "C" int MQ_SYM(mq_vlfuse_i2t_fwd)(const void* v_ln, const void* kf, const void* vo, long kv_bs, long kv_hs, long kv_ts, const float* bias,
                                 const int* kv_len, const void* out_bias, void* out, int B, int N, int T, int heads, int max_kv, float clamp,
                                 int variant, void* stream) {
  if (B <= 0 || N <= 0) return 0;
  if (T < 1 || T > 256 || heads < 1 || heads > VH) return -1;
  if (kv_bs <= 0 && kv_hs <= 0 && kv_ts <= 0) { kv_ts = VD; kv_hs = (long)T * VD; kv_bs = (long)heads * T * VD; }      // 0, 0, 0: contiguous
  constexpr long CH16 = 16 / (long)sizeof(half_t);                        // 16-byte chunks: every row start must be one
  if (kv_ts < VD || kv_ts * 256 >= (1l << 31) || (kv_bs | kv_hs | kv_ts) % CH16 || ((size_t)kf | (size_t)vo) % 16) return -6;
  I2TParams p;
  p.kv_bs = kv_bs; p.kv_hs = kv_hs; p.kv_ts = kv_ts;
  p.v = (const half_t*)v_ln; p.kf = (const half_t*)kf; p.vo = (const half_t*)vo; p.bias = bias; p.kv_len = kv_len;
  p.obias = (const half_t*)out_bias; p.out = (half_t*)out; p.B = B; p.N = N; p.T = T; p.H = heads; p.clamp = clamp;
  const int kv = (kv_len && max_kv > 0) ? min(max_kv, T) : T;
  const int nt = (kv + TK - 1) / TK;
  const int nbl = min(4, max(1, (kv - (nt - 1) * TK + 15) / 16));       // live 16-key blocks of the last tile
  hipStream_t st = (hipStream_t)stream;
#ifdef MQ_PRIMARY_UNIT
  if (variant >= 100 && nt == 3 && nbl == 1) {                             // ablation timings (see vlfuse_i2t_kernel)
    switch (variant - 100) {
      case 1: return launch_i2t<3, 1, 1, 1>(p, st);
      case 3: return launch_i2t<3, 1, 1, 3>(p, st);
      case 4: return launch_i2t<3, 1, 1, 4>(p, st);
      case 8: return launch_i2t<3, 1, 1, 8>(p, st);
      case 11: return launch_i2t<3, 1, 1, 11>(p, st);
      case 15: return launch_i2t<3, 1, 1, 15>(p, st);
      default: return -4;
    }
  }
#endif
  if (variant != 1 && nt == 3 && nbl <= 2)                                 // Q in registers for 129 .. 160 keys (variant 1: Q tile in LDS, A/B)
    return nbl == 1 ? launch_i2t<3, 1, 1, 0, true>(p, st) : launch_i2t<3, 1, 2, 0, true>(p, st);
#define MQ_I2T(NT_)                                                        \
  switch (nbl) {                                                           \
    case 1: return launch_i2t<NT_, 1, 1>(p, st);                           \
    case 2: return launch_i2t<NT_, 1, 2>(p, st);                           \
    case 3: return launch_i2t<NT_, 1, 3>(p, st);                           \
    default: return launch_i2t<NT_, 1, 4>(p, st);                          \
  }
  switch (nt) {
    case 1: MQ_I2T(1)
    case 2: MQ_I2T(2)
    case 3: MQ_I2T(3)
    default: MQ_I2T(4)
  }
#undef MQ_I2T
}

// ------------------------------------------------------------------------------------------------ text side
struct T2IParams {
  const half_t* kf;       // [B, 8, T, 256] queries = folded text keys; element (b, h, t, :) at b * kv_bs + h * kv_hs + t * kv_ts
  long kv_bs, kv_hs, kv_ts;
  const half_t* v;        // [B, N, 256] LN(v): keys AND values
  float* ws;              // [nsplit][B*8][T][WS_LD] fp32 partials: O (un-normalised), then m, l
  half_t* out;            // [B, T, 8*256]
  const int* kv_len;      // [B] or nullptr: text rows >= kv_len[b] are padding -> not computed, written as zeros
  const unsigned char* kmask;   // [B, kmask_bs] or nullptr: 1 = image token is padding (masked as a key); kmask_bs % 4 == 0, >= 64*ceil(N/64)
  long kmask_bs;
  int H;                  // heads, <= VH
  int B, N, T, nsplit;
  int wr;                 // rows per wave of the main kernel (16 QB): granularity at which all-padding rows are skipped
  int members;            // workgroups per (image, key split): ceil(heads * ceil(max live rows / wr) / waves per workgroup)
  float clamp;
};
namespace { constexpr int WS_LD = VD + 4; }               // 260 floats: rows stay 16-byte aligned

// ABL: ablation timings as for vlfuse_i2t_kernel (bit 0 no tile loads, 1 no LDS commits, 2 no softmax arithmetic, 3 no fragment reads / MFMAs)
// TDMA (round 6, 16-bit builds): the key / value tiles go global -> LDS by LDS-DMA (`buffer_load ... lds`, csrc/common.h) instead of through a
// register ring: no tile registers, no ds_write, no register wait in front of the commit.  The tile rows are PADDED in LDS (pitch 272 halfs), a
// copy piece is 1 KB of consecutive LDS bytes: lane l of piece c writes LDS chunk g = 64 c + l, i.e. (row g / 34, position g % 34) -- positions
// 32, 33 are the padding (they fetch the row's chunk 0; nobody reads them).  Rows past the image come back as zeros (buffer range check; the
// keys are masked by position anyway).  Tile pos + 1 is requested at the top of step pos into the buffer step pos - 1 read, and has the whole
// step to land.  The two buffers are handed to the loop as __restrict__ pointers (vl_scoped_tiles): without alias scopes hipcc makes every
// LDS read behind the copy wait for it.
template <class T, class F>
__device__ __forceinline__ void vl_scoped_tiles(T* __restrict__ b0, T* __restrict__ b1, F f) { f(b0, b1); }

template <int QB, bool KM, int ABL = 0, bool TDMA = false>
__global__ __launch_bounds__(2048 / (QB * 4)) void vlfuse_t2i_kernel(T2IParams p) {
  static_assert(!TDMA || (!MQ_VL_SPLIT && ABL == 0), "the LDS-copied tiles are the 16-bit builds' (the split-precise planes are computed while staging)");
  constexpr int NTH = 2048 / (QB * 4), WR = 16 * QB;        // threads per workgroup (512 / 256), query rows per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* tiles = (half_t*)smem;                           // [2][TK][KS]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  // XCD-aware order: the workgroups of one (image, key split) stream the SAME image tokens -> one XCD, adjacent
  const int members = p.members;
  const int seq = blockIdx.x >> 3;
  const int group = (seq / members) * 8 + (blockIdx.x & 7);
  if (group >= p.B * p.nsplit) return;
  const int wq = seq % members;
  const int b = group / p.nsplit, split = group % p.nsplit;
  // Work units = (head, block of WR text rows), LIVE blocks only, packed densely over the waves of the group's workgroups: padded
  // caption tokens are masked as keys everywhere downstream and the post-processor never reads their logits, so their rows of this
  // attention are dead (the merge writes zeros).  (Round 2 gave a workgroup 128 rows of one head: a 141-token caption -- 9 live blocks
  // per head -- took 2 x 8 workgroups per group, 8 of them with ONE live wave, each streaming every key tile: 3 passes of the chip
  // instead of 2.)  A wave beyond the last unit only helps to stage the key tiles.
  const int kv_rows = p.kv_len ? max(1, min(p.T, p.kv_len[b])) : p.T;
  const int nblk = (kv_rows + WR - 1) / WR, units = p.H * nblk;
  if (wq * (NTH / 64) >= units) return;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int unit = wq * (NTH / 64) + wv;
  const bool wave_live = unit < units;
  const int h = wave_live ? unit / nblk : 0;
  const int row0 = (wave_live ? unit % nblk : 0) * WR;

  const int ntiles = (p.N + TK - 1) / TK;
  const int tps = (ntiles + p.nsplit - 1) / p.nsplit;
  const int t0 = split * tps, t1 = min(ntiles, t0 + tps);
  const int nt = max(t1 - t0, 0);

  vfrag qf[QB][8];
  const half_t* qb_ = p.kf + (long)b * p.kv_bs + (long)h * p.kv_hs;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int row = min(row0 + qb * 16 + l15, p.T - 1);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[qb][kk] = vl_frag(*(const half8*)(qb_ + (row * (int)p.kv_ts + kk * 32 + lg * 8)));
  }
  float4_ o[16][QB];
#pragma unroll
  for (int db = 0; db < 16; ++db)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) o[db][qb] = (float4_){0.f, 0.f, 0.f, 0.f};
  float m[QB], lsum[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) { m[qb] = MQ_NEG_BIG; lsum[qb] = 0.f; }   // per query column (m replicated over lg, lsum a per-lane partial)

  const half_t* vb = p.v + (long)b * p.N * VD;
  // register prefetch ring: tiles pos + 1 .. pos + RS are in flight while tile pos is on the MFMAs.  The ablation run of round 3
  // (profiles/r03_call8_microbench_vlfuse.json) takes a third off the launch when the tile loads are removed -- but a third slot
  // (RS = 3) changed nothing (0.3202 vs 0.3218 ms, GPU call 9): it is the L2 -> CU ingest of 32 KB per tile and workgroup, not its
  // latency, that the loads cost.  Two slots.
  constexpr int RS = TDMA ? 0 : MQ_VL_SPLIT ? 1 : 2;       // (split-precise: one slot -- a tile's fp32 chunks are twice the registers)
  TileRegs<NTH> slot[RS > 0 ? RS : 1];
  // TDMA: pieces of a tile and this lane's source offsets (bytes inside a tile of 64 rows x 512 B)
  constexpr int NWV = NTH / 64, NPC = TK * KS * (int)sizeof(half_t) / 1024, PR = (NPC + NWV - 1) / NWV;
  static_assert(TK * KS * sizeof(half_t) % 1024 == 0, "a tile is a whole number of 1 KB pieces");
  int leoff[PR];
#pragma unroll
  for (int i = 0; i < PR; ++i) {
    const int gch = (wv + i * NWV) * 64 + lane, row = gch / (KS / 8), posn = gch % (KS / 8);
    leoff[i] = (row * VD + (posn < VD / 8 ? posn : 0) * 8) * (int)sizeof(half_t);
  }
  const mq_rsrc v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, p.N * VD * (int)sizeof(half_t), 0x00020000);
  auto dma_tile = [&](int pos, half_t* buf) __attribute__((always_inline)) {
    const int t = t0 + min(pos, max(nt - 1, 0));
    const int soff = __builtin_amdgcn_readfirstlane(t * TK * VD * (int)sizeof(half_t));
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      const int piece = wv + i * NWV;
      if (piece < NPC) lds_stage_16b_buf(v_rsrc, soff, leoff[i], buf + piece * 512);
    }
  };
  if constexpr (ABL != 0) {
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
      for (int i = 0; i < 2048 / NTH; ++i) slot[r].r[i] = zero8();
  }
  auto issue = [&](auto SLOT, int pos) {
    constexpr int sl = decltype(SLOT)::value;
    const int t = t0 + min(pos, max(nt - 1, 0));
    if constexpr (!(ABL & 1)) tile_issue(slot[sl], vb, t * TK, p.N - 1, tid);
  };
  const float cl = p.clamp > 0.f ? p.clamp : 3.0e38f;
  constexpr float THR = 8.0f;     // deferred rescale: O / l are rescaled only when a row max grows by more than THR
  auto body = [&](auto PAR, auto SL, int pos, half_t* cur, half_t* nxt) __attribute__((always_inline)) {   // PAR = pos % 2 (LDS buffer cur), SL = pos % RS (ring slot that held tile pos)
    constexpr int sl = decltype(SL)::value;
    if constexpr (TDMA) dma_tile(pos + 1, nxt);            // nxt was read in step pos - 1: free since that step's barrier
    else issue(SL, pos + RS);                              // the slot of tile pos was committed one step ago
    __builtin_amdgcn_sched_barrier(0);
    if (pos < nt && wave_live) {
      const half_t* tile = cur;
      float4_ s[4][QB];
      if constexpr (!(ABL & 8)) qk_tile<false, QB>(tile, qf, nullptr, s, l15, lg);
      else {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) s[nb][qb] = (float4_){0.f, 0.f, 0.f, 0.f};
      }
      const int key0 = (t0 + pos) * TK;
      unsigned km[4] = {0u, 0u, 0u, 0u};                   // padding flags of this lane's 4 x 4 keys (GroundingDINO batches)
      if constexpr (KM) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) km[nb] = *(const unsigned*)(p.kmask + (long)b * p.kmask_bs + key0 + nb * 16 + 4 * lg);
      }
      float mx[QB];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) mx[qb] = MQ_NEG_BIG;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool valid = key0 + nb * 16 + 4 * lg + r < p.N && (!KM || ((km[nb] >> (8 * r)) & 0xffu) == 0u);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            float v = __builtin_amdgcn_fmed3f(s[nb][qb][r], -cl, cl);         // one v_med3 (no clamp: cl = 3e38)
            v = valid ? v : MQ_NEG_BIG;
            s[nb][qb][r] = v;
            mx[qb] = fmaxf(mx[qb], v);
          }
        }
      bool grow = false;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        mx[qb] = fmaxf(mx[qb], __shfl_xor(mx[qb], 16));
        mx[qb] = fmaxf(mx[qb], __shfl_xor(mx[qb], 32));
        grow |= mx[qb] > m[qb] + THR;
      }
      if (__any(grow)) {                                   // wave-uniform, rare after the first tiles
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const float mnew = fmaxf(m[qb], mx[qb]);
          const float alpha = __expf(m[qb] - mnew);
          m[qb] = mnew;
          lsum[qb] *= alpha;
#pragma unroll
          for (int db = 0; db < 16; ++db)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[db][qb][r] *= alpha;
        }
      }
      half8 pf[2][QB];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e0 = (ABL & 4) ? s[2 * st][qb][r] : __expf(s[2 * st][qb][r] - m[qb]);          // <= e^THR; exact after the final 1/l
            const float e1 = (ABL & 4) ? s[2 * st + 1][qb][r] : __expf(s[2 * st + 1][qb][r] - m[qb]);
            lsum[qb] += e0 + e1;
            pf[st][qb][r] = (half_t)e0;
            pf[st][qb][4 + r] = (half_t)e1;
          }
      if constexpr (!(ABL & 8)) pv_tile(tile, pf, o, l15, lg);                       // the SAME tile: values = keys
      else {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[0][qb][r] += (float)pf[0][qb][r] + (float)pf[1][qb][4 + r];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TDMA) __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0), as an instruction hipcc's wait-count pass books: the copy has landed
    else if constexpr (!(ABL & 2)) tile_commit(slot[(sl + 1) % (RS > 0 ? RS : 1)], nxt, tid);
    __syncthreads();
  };
  if constexpr (TDMA) {
    dma_tile(0, tiles);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    vl_scoped_tiles(tiles, tiles + TILE, [&](half_t* b0, half_t* b1) __attribute__((always_inline)) {
      for (int pos = 0; pos < nt; pos += 2) {
        body(S0{}, S0{}, pos, b0, b1);
        if (pos + 1 < nt) body(S1{}, S0{}, pos + 1, b1, b0);             // workgroup-uniform: every wave runs the same number of steps
      }
    });
  } else {
  issue(S0{}, 0);
  if constexpr (RS >= 2) issue(S1{}, 1);
  if constexpr (RS == 3) issue(std::integral_constant<int, 2>{}, 2);
  tile_commit(slot[0], tiles, tid);
  __syncthreads();
  constexpr int U = RS == 3 ? 6 : 2;                       // lcm(2, RS) steps: every (buffer, slot) combination once
  for (int pos = 0; pos < nt; pos += U) {
    bool done = false;
    static_for_u<U>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      if (!done) {
        body(std::integral_constant<int, u & 1>{}, std::integral_constant<int, u % (RS > 0 ? RS : 1)>{}, pos + u, tiles + (u & 1) * TILE, tiles + ((u & 1) ^ 1) * TILE);
        done = pos + u + 1 >= nt;                          // workgroup-uniform: every wave runs the same number of steps
      }
    });
  }
  }

  // ---- partials -> workspace (O^T: lane owns 4 consecutive d of one query row -> one 16-byte store)
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float l = lsum[qb];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const int row = row0 + qb * 16 + l15;
    if (row < p.T && wave_live) {
      float* w = p.ws + (((long)split * p.B * p.H + (long)b * p.H + h) * p.T + row) * WS_LD;
#pragma unroll
      for (int db = 0; db < 16; ++db) *(float4_*)(w + db * 16 + 4 * lg) = o[db][qb];
      if (lg == 0) { w[VD] = m[qb]; w[VD + 1] = l; }
    }
  }
}

// merge the key-split partials: one wave per (b, h, text row); out[b, row, h*256 + d]
__global__ __launch_bounds__(256) void vlfuse_t2i_combine_kernel(T2IParams p) {
  const int lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long total = (long)p.B * p.H * p.T;
  if (gw >= total) return;
  const int row = gw % p.T;
  const int bh = gw / p.T, b = bh / p.H, h = bh % p.H;
  half_t* dst = p.out + ((long)b * p.T + row) * (p.H * VD) + h * VD + lane * 4;
  if (p.kv_len) {                                          // rows of skipped (all-padding) 16-row wave blocks: zeros
    const int kv = max(1, min(p.T, p.kv_len[b]));
    if ((row / p.wr) * p.wr >= kv) {
      *(half4*)dst = (half4){(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
      return;
    }
  }
  const long stride = total * WS_LD;
  const float* base = p.ws + gw * WS_LD;
  float mx = MQ_NEG_BIG;
  for (int s = 0; s < p.nsplit; ++s) mx = fmaxf(mx, base[s * stride + VD]);
  float l = 0.f;
  float4_ acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < p.nsplit; ++s) {
    const float* w = base + s * stride;
    const float f = __expf(w[VD] - mx);
    l += w[VD + 1] * f;
    const float4_ v = *(const float4_*)(w + lane * 4);
    acc += v * f;
  }
  const float inv = 1.f / l;
  half4 y;
#pragma unroll
  for (int j = 0; j < 4; ++j) y[j] = (half_t)(acc[j] * inv);
  *(half4*)dst = y;
}

#ifdef MQ_PRIMARY_UNIT
extern "C" long mq_vlfuse_t2i_workspace_bytes(int B, int T, int nsplit) {
  return (long)(nsplit < 1 ? 1 : nsplit) * B * VH * T * WS_LD * (long)sizeof(float);
}
#endif

// Text side of VLFuse (always through the split workspace + combine, nsplit >= 1).  See include/mqdet_hip.h.
extern "C" int MQ_SYM(mq_vlfuse_t2i_fwd)(const void* kf, long kv_bs, long kv_hs, long kv_ts, const void* v_ln, const int* kv_len,
                                 const unsigned char* key_mask, long key_mask_bs,
                                 void* workspace, void* out, int B, int N, int T, int heads, int nsplit, int max_kv, float clamp,
                                 int variant, void* stream) {
  if (B <= 0 || T <= 0) return 0;
  if (N < 1 || workspace == nullptr || heads < 1 || heads > VH) return -1;
  if (kv_bs <= 0 && kv_hs <= 0 && kv_ts <= 0) { kv_ts = VD; kv_hs = (long)T * VD; kv_bs = (long)heads * T * VD; }      // 0, 0, 0: contiguous
  if (kv_ts < VD || kv_ts * 256 >= (1l << 31) || (kv_bs | kv_hs | kv_ts) % (16 / (long)sizeof(half_t)) || (size_t)kf % 16) return -6;
  if (key_mask && ((key_mask_bs % 4) || key_mask_bs < (long)((N + TK - 1) / TK) * TK)) return -3;
  if (nsplit < 1) nsplit = 1;
  T2IParams p;
  p.kf = (const half_t*)kf; p.v = (const half_t*)v_ln; p.ws = (float*)workspace; p.out = (half_t*)out; p.kv_len = kv_len;
  p.kv_bs = kv_bs; p.kv_hs = kv_hs; p.kv_ts = kv_ts;
  p.B = B; p.N = N; p.T = T; p.H = heads; p.nsplit = nsplit; p.clamp = clamp; p.wr = 16 * vlfuse_qb();
  p.kmask = key_mask; p.kmask_bs = key_mask_bs;
  constexpr size_t smem = (size_t)2 * TILE * sizeof(half_t);
  static MqOncePerDevice attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)vlfuse_t2i_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)vlfuse_t2i_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)vlfuse_t2i_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)vlfuse_t2i_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set.done();
  }
  // key / value tiles by LDS-DMA (default since GPU call 20 of round 6: 259.9 -> 238.0 us per launch, +0.9 % end to end); MQ_VL_T2I_DMA=0: register ring
  static const bool t2i_dma = [] { const char* e = getenv("MQ_VL_T2I_DMA"); return !(e && e[0] == '0'); }();
  (void)t2i_dma;
  const int rows = (kv_len && max_kv > 0) ? min(max_kv, T) : T;          // host-known bound of the live text rows
  const int groups = B * nsplit, members = (heads * ((rows + p.wr - 1) / p.wr) + (128 / p.wr) - 1) / (128 / p.wr);
  p.members = members;
  const dim3 grid((unsigned)(8 * ((groups + 7) / 8) * members));
#ifdef MQ_PRIMARY_UNIT
  if (variant >= 100 && !key_mask) {                                       // ablation timings (tools/microbench.py; results are garbage)
    hipStream_t st = (hipStream_t)stream;
    switch (variant - 100) {
#define MQ_T2I_ABL(A_)                                                                                                              \
      case A_: {                                                                                                                    \
        hipError_t e = hipFuncSetAttribute((const void*)vlfuse_t2i_kernel<1, false, A_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        if (e != hipSuccess) return (int)e;                                                                                         \
        hipLaunchKernelGGL((vlfuse_t2i_kernel<1, false, A_>), grid, dim3(512), smem, st, p);                                        \
        break;                                                                                                                      \
      }
      MQ_T2I_ABL(1) MQ_T2I_ABL(3) MQ_T2I_ABL(4) MQ_T2I_ABL(8) MQ_T2I_ABL(11) MQ_T2I_ABL(15)
#undef MQ_T2I_ABL
      default: return -4;
    }
    MQ_CHECK_LAUNCH();
    return 0;
  }
#endif
  if (vlfuse_qb() == 1) {
    if (key_mask) hipLaunchKernelGGL((vlfuse_t2i_kernel<1, true>), grid, dim3(512), smem, (hipStream_t)stream, p);
#if !MQ_VL_SPLIT
    else if (t2i_dma && (long)N * VD * (long)sizeof(half_t) < (1l << 31)) {
      static MqOncePerDevice attr_dma;
      if (attr_dma.first()) {
        hipError_t e = hipFuncSetAttribute((const void*)vlfuse_t2i_kernel<1, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_dma.done();
      }
      hipLaunchKernelGGL((vlfuse_t2i_kernel<1, false, 0, true>), grid, dim3(512), smem, (hipStream_t)stream, p);
    }
#endif
    else hipLaunchKernelGGL((vlfuse_t2i_kernel<1, false>), grid, dim3(512), smem, (hipStream_t)stream, p);
  } else {
    if (key_mask) hipLaunchKernelGGL((vlfuse_t2i_kernel<2, true>), grid, dim3(256), smem, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((vlfuse_t2i_kernel<2, false>), grid, dim3(256), smem, (hipStream_t)stream, p);
  }
  MQ_CHECK_LAUNCH();
  const long total = (long)B * heads * T;
  hipLaunchKernelGGL(vlfuse_t2i_combine_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

MQ_NAMESPACE_END
