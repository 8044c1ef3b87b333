// mq_window_attn_fwd: Swin (shifted-)window multi-head self-attention for gfx950.
//
// Replaces, in one launch, the reference chain of swint.py:201-234 + :111-139 -- F.pad to a multiple
// of the window, torch.roll(-shift), window_partition, q*scale @ k^T, + relative-position bias,
// + SW-MSA mask (-100 across regions), softmax, @ v, window_reverse, roll(+shift), crop -- none of
// which moves data here: the roll / pad / partition are folded into the load and store addresses.
//
//   qkv      : [B, H, W, 3*C] fp16 = Linear(norm1(x)) on the UNPADDED tokens, C = heads*32
//   qkv_bias : [3*C] fp16 -- q/k/v of a pad token (pad happens after norm1, so a pad token is an exact
//              zero vector and its qkv row equals the bias; pad tokens DO act as keys, swint.py quirk 7)
//   rel_bias : [heads, NP, NP] fp32: relative_position_bias_table gathered by relative_position_index into rows = query,
//              cols = key, zero-padded from N = ws*ws to NP = 64 (N <= 64: Swin-T/S/B, window 7) or 160 (N <= 160:
//              Swin-L, window 12 -> 144 tokens); the kernel masks the padded keys itself
//   out      : [B, H, W, C] fp16 (attention output before `proj`)
// One wave per (window, head): N = ws*ws tokens padded to NP = 16 * NB (NB = 4 or 10 blocks of 16), head_dim 32.
//   S^T (NP keys x NP queries) = NB x NB mfma 16x16x32 with the operands SWAPPED (A = K rows, B = Q rows; K = head_dim = 32,
//     one MFMA per 16x16 block), fragments read straight from global memory (16 B per lane, each element used once).
//     A lane then owns one QUERY column (l15) and 4 keys (4*lg + r) per 16-key block: bias rows are 16-byte loads, the
//     softmax reduction is in-lane + 2 shuffles, and the normalised probabilities already ARE the P^T B-fragments of
//     the second MFMA -- P never goes through LDS;
//   O^T = V^T P^T: V is staged row-major in LDS ([key][32], one 16-byte store per lane and block) and read transposed
//     with ds_read_b64_tr_b16; a lane ends with 4 consecutive channels of one query -> 8-byte stores.
// (v1 computed S, wrote P and a scalar-transposed V through LDS, fetched the bias with 64 scalar loads per lane and
//  stored 2 bytes per lane: ~1 TB/s.  profiles/README.md.)
#include "common.h"

MQ_NAMESPACE_BEGIN


struct WinParams {
  const half_t* qkv; const half_t* qkv_bias; const float* rel_bias; half_t* out;
  int B, H, W, C, heads, ws, shift, Hp, Wp, nWx, nWy;
  float scale;
};

template <int NB>
__global__ __launch_bounds__(256) void window_attn_kernel(WinParams p) {
  constexpr int VP = 32 + 8;                               // V row pitch (halfs)
  constexpr int NP = 16 * NB;                              // padded window length
  static_assert(NB % 2 == 0, "PV walks the keys in steps of 32");
  __shared__ __attribute__((aligned(16))) half_t Vs_all[4][NP * VP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const long unit = (long)blockIdx.x * 4 + wave;          // (b, wy, wx, head)
  const long total = (long)p.B * p.nWy * p.nWx * p.heads;
  if (unit >= total) return;                               // wave-uniform; no block-level sync below
  const int head = unit % p.heads;
  long win = unit / p.heads;
  const int wx = win % p.nWx; win /= p.nWx;
  const int wy = win % p.nWy;
  const int b = win / p.nWy;
  const int N = p.ws * p.ws;
  half_t* Vs = Vs_all[wave];

  // window token i -> SW-MSA region id (img_mask of swint.py:570-588) in the shifted frame
  auto region_of = [&](int i) {
    const int ii = min(i, N - 1);
    const int ys = wy * p.ws + ii / p.ws, xs = wx * p.ws + ii % p.ws;
    const int ry = ys < p.Hp - p.ws ? 0 : (ys < p.Hp - p.shift ? 1 : 2);
    const int rx = xs < p.Wp - p.ws ? 0 : (xs < p.Wp - p.shift ? 1 : 2);
    return ry * 3 + rx;
  };

  // ---- the NB tokens this lane addresses (token = blk*16 + l15): K / Q fragments, V chunk, output pixel.
  // All 3*NB 16-byte loads are issued back to back BEFORE anything consumes them (addresses first, then loads, then the LDS
  // stores of V): written as one loop, the compiler kept the per-block order load -> wait -> ds_write, i.e. NB serialised
  // round trips to HBM / L2 per wave (ISA of v2: s_waitcnt vmcnt(2) after every third load).
  half8 qf[NB], kf[NB];
  long out_off[NB];                                        // element offset of the token's output row, -1: pad / beyond N
  int region_q[NB];
  {
    const half_t* rows[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      const int i = blk * 16 + l15, ii = min(i, N - 1);
      const int ys = wy * p.ws + ii / p.ws, xs = wx * p.ws + ii % p.ws;     // coords in the shifted frame
      int y = ys + p.shift; if (y >= p.Hp) y -= p.Hp;                     // shifted[ys] = x[(ys+shift) % Hp]
      int x = xs + p.shift; if (x >= p.Wp) x -= p.Wp;
      const bool real = (y < p.H) && (x < p.W);
      const long tok = ((long)b * p.H + y) * p.W + x;
      rows[blk] = (real ? p.qkv + tok * (3 * p.C) : p.qkv_bias) + head * 32 + lg * 8;
      out_off[blk] = (real && i < N) ? tok * p.C + head * 32 : -1;
      region_q[blk] = region_of(i);
    }
    half8 vv[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      vv[blk] = *(const half8*)(rows[blk] + 2 * p.C);
      qf[blk] = *(const half8*)(rows[blk]);
      kf[blk] = *(const half8*)(rows[blk] + p.C);
    }
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)                     // keys >= N hold a copy of key N-1: finite, weight exactly 0
      *(half8*)(Vs + (blk * 16 + l15) * VP + lg * 8) = vv[blk];
  }

  int region_k[NB];                                        // four 4-bit region ids per 16-key block
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    region_k[nb] = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) region_k[nb] |= region_of(nb * 16 + 4 * lg + r) << (4 * r);
  }
  const float* rel = p.rel_bias + (long)head * NP * NP;
  wave_lds_fence();                                        // V tile written by this wave's own lanes
  // one 16-query block at a time (keeps the kernel under 128 VGPRs -> 4 waves / SIMD for this load / store bound op)
#pragma unroll
  for (int qb = 0; qb < NB; ++qb) {
    if (qb * 16 >= N) break;                               // query blocks made of padding only (wave-uniform)
    // ---- S^T = K Q^T : s[nb], element r <-> key nb*16 + 4*lg + r, query qb*16 + l15
    float4_ s[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) s[nb] = mfma16(kf[nb], qf[qb], (float4_){0.f, 0.f, 0.f, 0.f});
    // the NB bias rows of this query block: all loads first (they were consumed one by one behind a wave-uniform branch on
    // `shift`, which cut the loop into basic blocks and put an s_waitcnt vmcnt(0) behind every load)
    const float* relq = rel + (qb * 16 + l15) * NP + 4 * lg;
    float4_ rb[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) rb[nb] = *(const float4_*)(relq + nb * 16);
    const float pen = p.shift > 0 ? -100.0f : 0.0f;         // SW-MSA penalty, applied branch-free
    float mx = MQ_NEG_BIG;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = s[nb][r] * p.scale + rb[nb][r];
        v += (((region_k[nb] >> (4 * r)) & 15) != region_q[qb]) ? pen : 0.0f;
        if (nb * 16 + 4 * lg + r >= N) v = MQ_NEG_BIG;
        s[nb][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[nb][r] - mx);
        s[nb][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
    half8 pf[NB / 2];                                      // P^T B-fragments of the 32-key steps
#pragma unroll
    for (int st = 0; st < NB / 2; ++st)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pf[st][r] = (half_t)(s[2 * st][r] * inv);
        pf[st][4 + r] = (half_t)(s[2 * st + 1][r] * inv);
      }
    // ---- O^T[db] = V^T P^T : A = transposed reads of the row-major V tile (k-slot (lg, j) <-> key 16*(j/4) + 4*lg + j%4)
    float4_ o[2] = {(float4_){0.f, 0.f, 0.f, 0.f}, (float4_){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int st = 0; st < NB / 2; ++st) {
      const half_t* base = Vs + (st * 32 + 4 * lg + (l15 >> 2)) * VP + (l15 & 3) * 4;
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const half4 lo = lds_read_tr16(base + db * 16);
        const half4 hi = lds_read_tr16(base + 16 * VP + db * 16);
        half8 a;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = lo[j]; a[4 + j] = hi[j]; }
        o[db] = mfma16(a, pf[st], o[db]);
      }
    }
    // O^T C layout: row = channel db*16 + 4*lg + r, col = query qb*16 + l15  ->  4 consecutive channels per lane
    if (out_off[qb] >= 0) {
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        half4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (half_t)o[db][r];
        *(half4*)(p.out + out_off[qb] + db * 16 + 4 * lg) = v;
      }
    }
  }
}

extern "C" int MQ_SYM(mq_window_attn_fwd)(const void* qkv, const void* qkv_bias, const float* rel_bias, void* out,
                                  int B, int H, int W, int C, int heads, int ws, int shift, void* stream) {
  if (B <= 0) return 0;
  if (C != heads * 32 || ws * ws > 160 || shift < 0 || shift >= ws) return -1;
  WinParams p;
  p.qkv = (const half_t*)qkv; p.qkv_bias = (const half_t*)qkv_bias; p.rel_bias = rel_bias; p.out = (half_t*)out;
  p.B = B; p.H = H; p.W = W; p.C = C; p.heads = heads; p.ws = ws; p.shift = shift;
  p.Hp = (H + ws - 1) / ws * ws; p.Wp = (W + ws - 1) / ws * ws;
  p.nWy = p.Hp / ws; p.nWx = p.Wp / ws;
  p.scale = 1.0f / sqrtf(32.0f);
  long total = (long)B * p.nWy * p.nWx * heads;
  if (ws * ws <= 64)
    hipLaunchKernelGGL(window_attn_kernel<4>, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
  else                                                     // Swin-L: window 12 -> 144 tokens padded to 160
    hipLaunchKernelGGL(window_attn_kernel<10>, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

// ---- mq_window_attn_qkv_fwd: the qkv projection INSIDE the window attention (round 3).
// At Swin stage 1 (C = 96, 537 600 tokens at B = 8) the qkv tensor is 310 MB per block: the library GEMM writes it (88 us, HBM bound)
// and mq_window_attn_fwd reads it back (144 us at 2.9 TB/s) -- 620 MB of the block's traffic for a tensor nobody else reads.  Here one
// wave owns one WINDOW for all its heads: the 64 x C normalised tokens are loaded once as MFMA fragments (A and B fragments of
// 16x16x32 have the same lane layout: token l15, channels 8 lg .. + 7), and per head
//     Q^T = Wq X^T,  K^T = Wk X^T   (A = weight rows from LDS, B = X fragments; accumulator rows = head channel 4 lg + r, cols = token l15)
//     V   = X Wv^T                  (A = X fragments, B = weight rows; accumulator rows = token, cols = channel)
// whose accumulators ARE, after rounding to 16 bits (the rounding point of the reference's qkv tensor), the fragments of
//     S^T = K Q^T                   (A = K, B = Q: a lane holds the 8 k-slots (d = 4 lg + t, 16 + 4 lg + t) of token l15 -- the same
//                                    permutation of d in both operands cancels in the contraction)
//     O^T = V^T P^T                 (A = V^T: row = channel l15, k-slots = keys 4 lg + t of two adjacent 16-token blocks -- the key
//                                    order of the P^T fragment the softmax leaves in registers, as in mq_window_attn_fwd)
// -- no LDS round trip for Q, K, V or P.  The 3C x C weight matrix lives in LDS for the life of the (persistent) workgroup
// (55 KB at C = 96).  Pad tokens (pad happens after norm1) are exact zero rows of X: their q / k / v equal the bias, as the
// reference's are.  Everything after the projections is mq_window_attn_fwd's code.  Algorithmic HBM bytes: x in, out out (4 C per token).
struct WinQkvParams {
  const half_t* x; const half_t* w; const half_t* bias; const float* rel_bias; half_t* out;
  int B, H, W, heads, ws, shift, Hp, Wp, nWx, nWy;
  long windows;
  float scale;
};

// STREAM (C = 192: the 221 KB of weights do not fit): the 96 weight rows of ONE head (q, k, v: 3 x 32 rows, 36 KB) are staged per head
// by LDS-DMA into a double buffer while the previous head computes; the four waves of a workgroup (four windows) walk the heads in
// step, one barrier per head.  Rows are unpadded (a DMA piece is 1 KB of consecutive LDS bytes); the 16-byte chunk c of row r sits at
// chunk position (c & ~7) | ((c ^ r) & 7): conflict-free for the fragment reads (16 rows x one chunk).  The X fragments (96 VGPRs at
// C = 192) leave room for one wave per SIMD only.
// Split-precise build: every fragment that feeds more than one MFMA is split ONCE into (hi, lo) fp16 x 8 -- the X fragments of the window (kept for
// all heads: the fp32 copy dies), the weight fragment of a (matrix, channel block, k-step), q / k / v of a head, the probabilities of a key step -- and
// the MFMAs are mfma16_split: nothing is split twice, no fp32 copy stays live beside its split form.
#if defined(MQ_F32) && !defined(MQ_F32_EXACT)
#define MQ_WQ_SPLIT 1
typedef mq_split8 wq_frag;
#define WQ_F(x) mq_split(x)
#define WQ_MFMA(a, b, c) mfma16_split((a), (b), (c))
#else
#define MQ_WQ_SPLIT 0
typedef half8 wq_frag;
#define WQ_F(x) (x)
#define WQ_MFMA(a, b, c) mfma16((a), (b), (c))
#endif
// (split-precise build: the fp32 X fragments of a window and their split copies are twice the registers -- one workgroup per CU there: 256 VGPRs + 238
// AGPRs and no scratch at C = 96 instead of 178 spilled VGPRs, 2.0 GB of scratch traffic per launch in profiles/r06_pmc_traffic_split.json)
#if defined(MQ_F32)
#define MQ_WQKV_WG_PER_CU(STREAM_) 1
#else
#define MQ_WQKV_WG_PER_CU(STREAM_) ((STREAM_) ? 1 : 2)
#endif
template <int C, bool STREAM>
__global__ __launch_bounds__(256, MQ_WQKV_WG_PER_CU(STREAM)) void window_attn_qkv_kernel(WinQkvParams p) {
  constexpr int NB = 4, NP = 64, KS = C / 32, HEADS = C / 32, WP = STREAM ? C : C + 8;       // weight row pitch (halfs)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* Ws = (half_t*)smem;                              // resident: [3C][WP]; STREAM: [2][96][C]
  half_t* Bsm = Ws + (STREAM ? 2 * 96 * C : 3 * C * WP);   // [3C] qkv bias (a global load per use would be waited for on the spot)
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const mq_rsrc rs_w = mq_raw_buffer(p.w);
  // STREAM: the pieces of head h (36 x 1 KB = 96 rows x C halfs, 9 per wave) -> buffer `buf`
  auto stage_head = [&](int h, int buf) __attribute__((always_inline)) {
    if constexpr (STREAM) {
      constexpr int PIECES = 96 * C / 512;
#pragma unroll
      for (int i = 0; i < PIECES / 4; ++i) {
        const int pc = wave + 4 * i;
        const int off = pc * 512 + lane * 8, row = off / C, posn = (off % C) >> 3;
        const int c = (posn & ~7) | ((posn ^ row) & 7);
        half_t* dst = Ws + buf * 96 * C + pc * 512;
        // (MUBUF `buffer_load ... lds`, csrc/common.h: head offset in the scalar offset, this lane's swizzled source in the vector offset)
        lds_stage_frag8_buf(rs_w, p.w, h * 32 * C, ((row >> 5) * C + (row & 31)) * C + c * 8, dst, lane);
      }
    }
  };
  for (int c = tid; c < 3 * C / 8; c += 256) *(half8*)(Bsm + c * 8) = *(const half8*)(p.bias + c * 8);
  if constexpr (!STREAM) {
    for (int c = tid; c < 3 * C * (C / 8); c += 256) {
      const int row = c / (C / 8), ch = c % (C / 8);
      *(half8*)(Ws + row * WP + ch * 8) = *(const half8*)(p.w + (long)row * C + ch * 8);
    }
    __syncthreads();
  } else {
    stage_head(0, 0);
  }
  const int N = p.ws * p.ws;
  int seq = 0;                                             // STREAM: heads processed so far by this workgroup (buffer parity)

  for (long base = (long)blockIdx.x * 4; base < p.windows; base += (long)gridDim.x * 4) {      // workgroup-uniform trip count
    const bool active = base + wave < p.windows;           // an idle wave of the last group repeats the last window without storing
    const long win = active ? base + wave : p.windows - 1;
    long t = win;
    const int wx = t % p.nWx; t /= p.nWx;
    const int wy = t % p.nWy;
    const int b = t / p.nWy;
    auto region_of = [&](int i) {
      const int ii = min(i, N - 1);
      const int ys = wy * p.ws + ii / p.ws, xs = wx * p.ws + ii % p.ws;
      const int ry = ys < p.Hp - p.ws ? 0 : (ys < p.Hp - p.shift ? 1 : 2);
      const int rx = xs < p.Wp - p.ws ? 0 : (xs < p.Wp - p.shift ? 1 : 2);
      return ry * 3 + rx;
    };
    // ---- X fragments: token blk * 16 + l15, channels 32 ks + 8 lg .. + 7; tokens >= N copy token N - 1 (masked as keys, not stored)
    wq_frag xf[NB][KS];
    int out_off[NB];                                       // element offset of the token's row (B * H * W * C < 2^31: checked by the host), -1: none
    int region_q[NB];
    {
      int rows[NB];
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        const int i = blk * 16 + l15, ii = min(i, N - 1);
        const int ys = wy * p.ws + ii / p.ws, xs = wx * p.ws + ii % p.ws;
        int y = ys + p.shift; if (y >= p.Hp) y -= p.Hp;
        int x = xs + p.shift; if (x >= p.Wp) x -= p.Wp;
        const bool real = (y < p.H) && (x < p.W);
        const int tok = (b * p.H + y) * p.W + x;
        rows[blk] = real ? tok * C + lg * 8 : -1;                  // pad token: X row = 0
        out_off[blk] = (real && i < N && active) ? tok * C : -1;
        region_q[blk] = region_of(i);
      }
#if MQ_WQ_SPLIT
      half8 xr[NB][KS];
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xr[blk][ks] = *(const half8*)(p.x + max(rows[blk], 0) + ks * 32);       // unconditional, in flight together
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[blk][ks] = mq_split(rows[blk] >= 0 ? xr[blk][ks] : zero8());
#else
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[blk][ks] = *(const half8*)(p.x + max(rows[blk], 0) + ks * 32);       // unconditional, in flight together
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[blk][ks] = rows[blk] >= 0 ? xf[blk][ks] : zero8();
#endif
    }
    int region_k[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      region_k[nb] = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) region_k[nb] |= region_of(nb * 16 + 4 * lg + r) << (4 * r);
    }

#pragma unroll 1
    for (int head = 0; head < HEADS; ++head) {
      if constexpr (STREAM) {
        // this head's pieces (issued one head ago) have landed for every wave, and every wave is done with the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool last = head + 1 == HEADS && base + (long)gridDim.x * 4 >= p.windows;
        if (!last) stage_head((head + 1) % HEADS, (seq + 1) & 1);
      }
      const half_t* Wb = STREAM ? Ws + (seq & 1) * 96 * C : Ws;
      // ---- projections of this head.  m = 0 (q), 1 (k): transposed; 2 (v): plain.  bias: row (d) for q / k, column (d) for v
      constexpr int WQ_RD = MQ_WQ_SPLIT ? 1 : 2;            // depth of the weight-fragment ring of the projections
      wq_frag qf[NB], kf[NB], vf[NB / 2][2];                // vf[st][db]: V^T A-fragment of 32-key step st, channel block db
      {
        // A / B fragment of weight rows (matrix m, channel block db) for k-step ks: row l15, channels 32 ks + 8 lg .. + 7
        auto wfrag = [&](int m, int db, int ks) __attribute__((always_inline)) -> half8 {
          if constexpr (STREAM) {
            const int row = m * 32 + db * 16 + l15, c = ks * 4 + lg;
            return *(const half8*)(Wb + row * C + (((c & ~7) | ((c ^ row) & 7)) << 3));
          } else {
            return *(const half8*)(Wb + (m * C + head * 32 + db * 16 + l15) * WP + lg * 8 + ks * 32);
          }
        };
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          float4_ acc[2][NB];
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            const half4 bv = *(const half4*)(Bsm + m * C + head * 32 + db * 16 + 4 * lg);
#pragma unroll
            for (int tb = 0; tb < NB; ++tb)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[db][tb][r] = (float)bv[r];
          }
          // the weight fragments come through a ring WQ_RD reads deep: read -> wait -> four MFMAs per fragment (the ISA of round 5) exposes the
          // LDS latency once per 64 matrix cycles, with one or two waves per SIMD to cover it
          half8 wring[WQ_RD];
#pragma unroll
          for (int i = 0; i < WQ_RD; ++i) wring[i] = wfrag(m, i & 1, i >> 1);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
              const int i = ks * 2 + db;
              const wq_frag wf = WQ_F(wring[i % WQ_RD]);
              if (i + WQ_RD < 2 * KS) wring[i % WQ_RD] = wfrag(m, (i + WQ_RD) & 1, (i + WQ_RD) >> 1);
#pragma unroll
              for (int tb = 0; tb < NB; ++tb) acc[db][tb] = WQ_MFMA(wf, xf[tb][ks], acc[db][tb]);
              if constexpr (!MQ_WQ_SPLIT) __builtin_amdgcn_sched_barrier(0);      // source order = issue order (hipcc sinks the ring's reads to their uses)
            }
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) {
            half8 f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { f[r] = (half_t)acc[0][tb][r]; f[4 + r] = (half_t)acc[1][tb][r]; }
            if (m == 0) qf[tb] = WQ_F(f); else kf[tb] = WQ_F(f);
          }
          __builtin_amdgcn_sched_barrier(0);                 // one projection at a time: their accumulators must not be live together
        }
        float4_ acc[NB][2];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const float bv = (float)Bsm[2 * C + head * 32 + db * 16 + l15];
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) acc[tb][db] = (float4_){bv, bv, bv, bv};
        }
        half8 wring[WQ_RD];
#pragma unroll
        for (int i = 0; i < WQ_RD; ++i) wring[i] = wfrag(2, i & 1, i >> 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            const int i = ks * 2 + db;
            const wq_frag wf = WQ_F(wring[i % WQ_RD]);
            if (i + WQ_RD < 2 * KS) wring[i % WQ_RD] = wfrag(2, (i + WQ_RD) & 1, (i + WQ_RD) >> 1);
#pragma unroll
            for (int tb = 0; tb < NB; ++tb) acc[tb][db] = WQ_MFMA(xf[tb][ks], wf, acc[tb][db]);
            if constexpr (!MQ_WQ_SPLIT) __builtin_amdgcn_sched_barrier(0);
          }
#if MQ_WQ_SPLIT
#pragma unroll
        for (int st = 0; st < NB / 2; ++st)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            half8 v8;
#pragma unroll
            for (int r = 0; r < 4; ++r) { v8[r] = (half_t)acc[2 * st][db][r]; v8[4 + r] = (half_t)acc[2 * st + 1][db][r]; }
            vf[st][db] = mq_split(v8);
          }
#else
#pragma unroll
        for (int st = 0; st < NB / 2; ++st)
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 4; ++r) { vf[st][db][r] = (half_t)acc[2 * st][db][r]; vf[st][db][4 + r] = (half_t)acc[2 * st + 1][db][r]; }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      const float* rel = p.rel_bias + (long)head * NP * NP;
#pragma unroll
      for (int qb = 0; qb < NB; ++qb) {
        if (qb * 16 >= N) break;
        float4_ s[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) s[nb] = WQ_MFMA(kf[nb], qf[qb], ((float4_){0.f, 0.f, 0.f, 0.f}));
        const float* relq = rel + (qb * 16 + l15) * NP + 4 * lg;
        float4_ rb[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) rb[nb] = *(const float4_*)(relq + nb * 16);
        const float pen = p.shift > 0 ? -100.0f : 0.0f;
        float mx = MQ_NEG_BIG;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = s[nb][r] * p.scale + rb[nb][r];
            v += (((region_k[nb] >> (4 * r)) & 15) != region_q[qb]) ? pen : 0.0f;
            if (nb * 16 + 4 * lg + r >= N) v = MQ_NEG_BIG;
            s[nb][r] = v;
            mx = fmaxf(mx, v);
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __expf(s[nb][r] - mx);
            s[nb][r] = e;
            sum += e;
          }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.f / sum;
        float4_ o[2] = {(float4_){0.f, 0.f, 0.f, 0.f}, (float4_){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int st = 0; st < NB / 2; ++st) {
          half8 pf;
#pragma unroll
          for (int r = 0; r < 4; ++r) { pf[r] = (half_t)(s[2 * st][r] * inv); pf[4 + r] = (half_t)(s[2 * st + 1][r] * inv); }
          const wq_frag ps = WQ_F(pf);
#pragma unroll
          for (int db = 0; db < 2; ++db) o[db] = WQ_MFMA(vf[st][db], ps, o[db]);
        }
        if (out_off[qb] >= 0) {
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            half4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (half_t)o[db][r];
            *(half4*)(p.out + out_off[qb] + head * 32 + db * 16 + 4 * lg) = v;
          }
        }
        __builtin_amdgcn_sched_barrier(0);                   // one query block at a time (bias rows of the next are not hoisted)
      }
      ++seq;
    }
  }
}

template <int C, bool STREAM>
static int launch_window_attn_qkv(const WinQkvParams& p, hipStream_t s) {
  constexpr size_t smem = (STREAM ? (size_t)2 * 96 * C : (size_t)3 * C * (C + 8)) * sizeof(half_t) + (size_t)3 * C * sizeof(half_t);
  static MqOncePerDevice attr;
  if (attr.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)window_attn_qkv_kernel<C, STREAM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr.done();
  }
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const long wgs_needed = (p.windows + 3) / 4, slots = (long)cus * MQ_WQKV_WG_PER_CU(STREAM);  // persistent workgroups: as many as the chip holds
  const long wgs = wgs_needed < slots ? wgs_needed : slots;
  hipLaunchKernelGGL((window_attn_qkv_kernel<C, STREAM>), dim3((unsigned)wgs), dim3(256), smem, s, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

// x [B,H,W,C] 16-bit = norm1(x) on the UNPADDED tokens, w [3C, C] / bias [3C] = attn.qkv (nn.Linear layout), rel_bias as
// mq_window_attn_fwd, out [B,H,W,C].  C = heads * 32 in {96, 192}; windows of at most 64 tokens.  Returns -1 otherwise (callers use the
// GEMM + mq_window_attn_fwd pair).  C = 192 streams the weights per head (see the kernel).
extern "C" int MQ_SYM(mq_window_attn_qkv_fwd)(const void* x, const void* w, const void* bias, const float* rel_bias, void* out,
                                      int B, int H, int W, int C, int heads, int ws, int shift, void* stream) {
  if (B <= 0) return 0;
  if (C != heads * 32 || ws * ws > 64 || shift < 0 || shift >= ws || (C != 96 && C != 192) || (long)B * H * W * C >= (1L << 31)) return -1;
  WinQkvParams p;
  p.x = (const half_t*)x; p.w = (const half_t*)w; p.bias = (const half_t*)bias; p.rel_bias = rel_bias; p.out = (half_t*)out;
  p.B = B; p.H = H; p.W = W; p.heads = heads; p.ws = ws; p.shift = shift;
  p.Hp = (H + ws - 1) / ws * ws; p.Wp = (W + ws - 1) / ws * ws;
  p.nWy = p.Hp / ws; p.nWx = p.Wp / ws;
  p.windows = (long)B * p.nWy * p.nWx;
  p.scale = 1.0f / sqrtf(32.0f);
  return C == 96 ? launch_window_attn_qkv<96, false>(p, (hipStream_t)stream) : launch_window_attn_qkv<192, true>(p, (hipStream_t)stream);
}

MQ_NAMESPACE_END
