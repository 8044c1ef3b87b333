// mq_bert_attn_qkv_fwd: the attention half of a BERT layer in ONE launch -- the q | k | v projection of a (batch item, head) AND its
// attention (HF BertSelfAttention; the clamped copy of the VLDyHead fusion layers, rpn/modeling_bert.py:119-170) -- gfx950, round 5.
//
// Why (north-star "fused GCP + BERT attention"): mq_attn_text_fwd is memory- and launch-bound -- at B = 64 one launch moves
// >= 100 MB of Q / K / V / O for 9.7 GFLOP (an HBM roofline of 31 % of the MFMA peak) behind a library GEMM that wrote those 75 MB.  Here the
// qkv tensor never exists: a workgroup (8 waves) owns one (b, h):
//   phase 1  [T x 192] = X_b [T x C] . W_h^T  (W_h = the head's 64 rows of Wq, Wk and Wv): X_b is staged through LDS in 64-wide k-chunks
//            (double buffer, register prefetch, one barrier per chunk) and shared by all waves; a wave owns output columns 48 wn .. 48 wn + 47 of
//            every second token block, so its weight fragments go global (L2) -> registers, prefetched one chunk ahead, never through LDS.
//            Per chunk and wave: 2 x (<= 16 A-fragment reads + 3 x <= 16 MFMAs); accumulators 3 x NBM tiles.
//   hand-over  + bias, rounded to the operand type (the rounding point of the reference's q / k / v tensors), written row-major [token][64]
//            into three LDS tiles that alias the X stages;
//   phase 2  per 16-query block (blocks dealt round-robin to the waves): S^T = K Q^T with all logits of a query in registers, exact two-pass
//            softmax in the exp2 domain, O^T = V^T P^T with V read transposed out of LDS (ds_read_tr16_b64) -- the loop body of mq_attn_text_fwd.
// Text tokens are few (T <= 256, after live-row compaction T = 16 ceil(caption / 16)), so one workgroup holds a whole (b, h).
#include "common.h"
#include <cstdlib>
#include <type_traits>

MQ_NAMESPACE_BEGIN

namespace {
constexpr int BA_D = 64, BA_KS = BA_D + 16;        // head width; row pitch (elements) of the Q / K / V tiles: conflict-free b128 and transposed reads
constexpr int BA_BK = 64, BA_XP = BA_BK + 16;      // k-chunk of the projection; row pitch of an X stage
constexpr int BA_C = 768;                          // hidden width (BERT-base; compile time: the chunk loop is unrolled completely)
constexpr float BA_LOG2E = 1.4426950408889634f;

template <int I, int N, class F>
__device__ __forceinline__ void ba_static_for_impl(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ba_static_for_impl<I + 1, N>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void ba_static_for(F f) { ba_static_for_impl<0, N>(f); }
}  // namespace

struct BertAttnParams {
  const half_t* x;                // [B, T, C] hidden states (operand type), row stride x_rs, batch stride x_bs
  const half_t* w;                // [3 C, C]: rows q | k | v (the layer's fused projection weight) in MFMA B-fragment order [3 C / 16][C / 32][64][8]
  const half_t* bias;             // [3 C]
  half_t* o;                      // [B, T, C] context (heads concatenated)
  const float* key_bias;          // (b, j) at key_bias + b * bias_bs + j, or nullptr; <= -1e29 marks a masked key
  const int* kv_len;              // [B] or nullptr: 16-key blocks at and beyond kv_len[b] are skipped
  int B, T, C, H;
  long x_bs, x_rs, o_bs, o_rs, bias_bs;
  float scale, clamp;
  int nblk_cap;                   // 16-token blocks the LDS tiles hold (host: ceil(T / 16) <= NBM)
};

// NBM = 16-token blocks a workgroup can hold (compile time: 10 -> T <= 160, 16 -> T <= 256).
// EIGHT waves (GPU call 2 of round 5: the four-wave version ran one workgroup in 37 us -- 7x its MFMA time -- with one or two waves per SIMD
// nothing covered the L2 round trip of a chunk's operands or the dependent QK^T -> softmax -> PV chain of a query block): wave = (wn, wm),
// wn = column slice as above, wm = which half of the token blocks (blocks wm, wm + 2, ...) -- half the accumulators per wave, twice the
// waves per SIMD; phase 2 deals the query blocks to eight waves (two rounds for a 141-token caption instead of three).
// (Two workgroups per CU -- <= 128 VGPRs -- spilled ~60 dwords inside the chunk loop and ran 4x slower: GPU call 3 of round 5; one per CU.)
template <int NBM, bool CLAMP>
__global__ __launch_bounds__(512, 2) void bert_attn_qkv_kernel(BertAttnParams p) {
  constexpr int D = BA_D, KS = BA_KS, BK = BA_BK, XP = BA_XP, NT = 512, NMB = NBM / 2;
  constexpr int NX = (NBM * 16 * (BK / 8) + NT - 1) / NT;        // 16-byte chunks of an X stage per thread
  static_assert(NBM % 2 == 0, "two 16-key blocks per step of the P V product");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cap = p.nblk_cap, rows = 32 * ((cap + 1) >> 1);        // tile rows (a whole number of 32-key steps of the P V product)
  half_t* Xs = (half_t*)smem;                        // phase 1: [2][rows][XP]
  half_t* Qs = (half_t*)smem;                        // phase 2: [rows][KS] x 3 (aliases the X stages: 3 KS >= 2 XP)
  half_t* Ks = Qs + rows * KS;
  half_t* Vs = Ks + rows * KS;
  float* Bias_s = (float*)(Vs + rows * KS);          // [rows] log2(e) x key bias; masked / out-of-range keys: -1e30 (CLAMP: 0)
  float* Kmask_s = Bias_s + rows;                    // [rows] CLAMP only: 0, or -1e30 for masked / out-of-range keys

  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave & 3, wm = wave >> 2;
  const int h = blockIdx.x % p.H, b = blockIdx.x / p.H;
  const int T = p.T;
  constexpr int C = BA_C;
  const int mblk = (T + 15) >> 4;                    // 16-token blocks of this launch (<= cap)
  const int nk_eff = p.kv_len ? max(1, min(T, p.kv_len[b])) : T;
  const int nblk = min((nk_eff + 15) >> 4, mblk);    // 16-key blocks visited
  const int nst = (nblk + 1) >> 1;
  const half_t* X = p.x + (long)b * p.x_bs;

  for (int j = tid; j < rows; j += NT) {
    float kb = MQ_NEG_BIG;
    if (j < T) kb = p.key_bias ? p.key_bias[(long)b * p.bias_bs + j] : 0.f;
    const bool masked = kb < -1.0e29f;
    if constexpr (CLAMP) {
      Bias_s[j] = masked ? 0.f : kb * BA_LOG2E;
      Kmask_s[j] = masked ? MQ_NEG_BIG : 0.f;
    } else {
      Bias_s[j] = masked ? MQ_NEG_BIG : kb * BA_LOG2E;
    }
  }

  // ================================================================ phase 1: the projection
  // this wave's three 16-column blocks of the head's [q | k | v] = 192 columns: block cb = 3 wn + j is columns 16 (cb & 3) .. of part cb >> 2
  const half_t* wrow[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int cb = 3 * wn + j;
    wrow[j] = p.w + ((long)(((cb >> 2) * C + h * D) / 16 + (cb & 3)) * (C / 32) * 64 + lane) * 8;      // fragment-order weights: tile, k-step 0, lane
  }
  float4_ acc[NMB][3];                               // token blocks wm, wm + 2, ...
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[mb][j] = (float4_){0.f, 0.f, 0.f, 0.f};
  // Operands of chunk k + 2 are requested while chunk k is multiplied (GPU call 3 of round 5: with a distance of ONE chunk a chunk cost an L2
  // round trip, ~0.7 us, against 0.4 us of MFMA work): weight fragments in three register sets, X rows one more chunk in registers before
  // they go to their LDS stage (two stages suffice: stage (k + 1) & 1 was last read in chunk k - 1, behind a barrier).  The chunk loop is
  // unrolled completely (C = 768: 12 chunks) -- straight-line code lets the compiler count its vmcnt waits exactly; across a loop back edge
  // it waits for everything.
  half8 xr[2][NX];
  half8 wf[3][3][2];
  const int xchunks = mblk * 16 * (BK / 8);
  auto load_x = [&](half8 (&x)[NX], int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int c = tid + i * NT, r = c >> 3, ch = c & 7;
      if (c < xchunks) x[i] = *(const half8*)(X + (long)min(r, T - 1) * p.x_rs + ks * BK + ch * 8);
    }
  };
  auto store_x = [&](const half8 (&x)[NX], int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int c = tid + i * NT, r = c >> 3, ch = c & 7;
      if (c < xchunks) *(half8*)(Xs + ((long)buf * rows + r) * XP + ch * 8) = x[i];
    }
  };
  auto load_w = [&](half8 (&w)[3][2], int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) w[j][kk] = *(const half8*)(wrow[j] + (ks * (BK / 32) + kk) * (64 * 8));
  };
  auto gemm_chunk = [&](int buf, const half8 (&w)[3][2]) __attribute__((always_inline)) {
    const half_t* xt = Xs + (long)buf * rows * XP + (wm * 16 + l15) * XP + lg * 8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) {
        if (2 * mb + wm < mblk) {
          const half8 af = *(const half8*)(xt + mb * 32 * XP + kk * 32);
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[mb][j] = mfma16(af, w[j][kk], acc[mb][j]);
        }
      }
    }
  };
  constexpr int NSTEPS = BA_C / BK;
  load_x(xr[0], 0);
  load_w(wf[0], 0);
  load_x(xr[1], 1);
  load_w(wf[1], 1);
  store_x(xr[0], 0);
  __syncthreads();
  ba_static_for<NSTEPS>([&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k + 2 < NSTEPS) {
      load_x(xr[k & 1], k + 2);                      // (xr[k & 1] held chunk k: stored in chunk k - 1)
      load_w(wf[(k + 2) % 3], k + 2);
    }
    gemm_chunk(k & 1, wf[k % 3]);
    if constexpr (k + 1 < NSTEPS) store_x(xr[(k + 1) & 1], (k + 1) & 1);
    __syncthreads();                                 // (last chunk: every wave is done with the X stages before the tiles overwrite them)
  });

  // ================================================================ hand-over: + bias, one rounding, row-major tiles
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int cb = 3 * wn + j, part = cb >> 2, col = (cb & 3) * 16 + l15;
    const float bv = (float)p.bias[(long)part * C + h * D + col];
    half_t* tile = (part == 0 ? Qs : part == 1 ? Ks : Vs) + col;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      if (2 * mb + wm < mblk) {
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[((2 * mb + wm) * 16 + 4 * lg + r) * KS] = (half_t)(acc[mb][j][r] + bv);
      }
    }
  }
  // value rows behind the last token block that the last 32-key step still reads (its probabilities are exactly 0): finite data
  for (int c = tid; c < (rows - 16 * mblk) * (D / 8); c += NT) *(half8*)(Vs + (16 * mblk + c / (D / 8)) * KS + (c % (D / 8)) * 8) = zero8();
  __syncthreads();

  // ================================================================ phase 2: attention, one 16-query block at a time
  const float sc2 = p.scale * BA_LOG2E, cl2 = p.clamp * BA_LOG2E;
  half_t* O = p.o + (long)b * p.o_bs + h * D;
  for (int qblk = wave; qblk < mblk; qblk += 8) {
    half8 qf[D / 32];
#pragma unroll
    for (int kk = 0; kk < D / 32; ++kk) qf[kk] = *(const half8*)(Qs + (qblk * 16 + l15) * KS + kk * 32 + lg * 8);
    float4_ s[NBM];
    float mx = MQ_NEG_BIG;
#pragma unroll
    for (int nb = 0; nb < NBM; ++nb) {
      if (nb < nblk) {
        s[nb] = (float4_){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < D / 32; ++kk) {
          const half8 kf = *(const half8*)(Ks + (nb * 16 + l15) * KS + kk * 32 + lg * 8);
          s[nb] = mfma16(kf, qf[kk], s[nb]);
        }
        const float4_ kb4 = *(const float4_*)(Bias_s + nb * 16 + lg * 4);
        float4_ km4 = (float4_){0.f, 0.f, 0.f, 0.f};
        if constexpr (CLAMP) km4 = *(const float4_*)(Kmask_s + nb * 16 + lg * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = __builtin_fmaf(s[nb][r], sc2, kb4[r]);
          if constexpr (CLAMP) v = __builtin_amdgcn_fmed3f(v, -cl2, cl2) + km4[r];
          s[nb][r] = v;
          mx = fmaxf(mx, v);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float lsum = 0.f;
    float4_ o[D / 16];
#pragma unroll
    for (int db = 0; db < D / 16; ++db) o[db] = (float4_){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < NBM / 2; ++st) {
      if (st < nst) {
        half8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p0 = __builtin_amdgcn_exp2f(s[2 * st][r] - mx);
          const float p1 = (2 * st + 1 < nblk) ? __builtin_amdgcn_exp2f(s[2 * st + 1][r] - mx) : 0.f;
          lsum += p0 + p1;
          pf[r] = (half_t)p0;
          pf[4 + r] = (half_t)p1;
        }
        const half_t* base = Vs + (st * 32 + 4 * lg + (l15 >> 2)) * KS + (l15 & 3) * 4;
#pragma unroll
        for (int db = 0; db < D / 16; ++db) {
          const half4 lo = lds_read_tr16(base + db * 16);
          const half4 hi = lds_read_tr16(base + 16 * KS + db * 16);
          half8 a;
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[j] = lo[j]; a[4 + j] = hi[j]; }
          o[db] = mfma16(a, pf, o[db]);
        }
      }
    }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    const float inv = 1.f / lsum;
    const int row = qblk * 16 + l15;
    if (row < T) {
#pragma unroll
      for (int db = 0; db < D / 16; ++db) {
        half4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (half_t)(o[db][r] * inv);
        *(half4*)(O + (long)row * p.o_rs + db * 16 + lg * 4) = v;
      }
    }
  }
}

template <int NBM, bool CLAMP>
static int launch_bert_attn(const BertAttnParams& p, hipStream_t stream) {
  auto bytes = [](int cap) {
    const size_t rows = 32 * (size_t)((cap + 1) >> 1);
    return 3 * rows * BA_KS * sizeof(half_t) + 2 * rows * sizeof(float);
  };
  static MqOncePerDevice attr;
  if (attr.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)bert_attn_qkv_kernel<NBM, CLAMP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes(NBM));
    if (e != hipSuccess) return (int)e;
    attr.done();
  }
  hipLaunchKernelGGL((bert_attn_qkv_kernel<NBM, CLAMP>), dim3((unsigned)(p.B * p.H)), dim3(512), bytes(p.nblk_cap), stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

// x [B, T, C] operand type (row stride x_rs, batch stride x_bs, elements; % 8), w [3 C, C] = the layer's q | k | v projection weight in MFMA
// B-fragment order (mqdet_hip.h; one wave instruction then reads 1 KiB of consecutive bytes -- row-major fragments stream 3.7x slower, see
// gcp_fused.hip), bias [3 C],
// o [B, T, C] (o_rs % 4); key_bias fp32 (b, j) at key_bias + b * bias_bs + j or NULL; kv_len [B] int32 or NULL (keys at and beyond it are
// skipped in whole 16-key blocks; the caller's key_bias masks the rest).  C = 768 = 64 H (BERT-base), T <= 256.  clamp > 0: the +-clamp of the
// VLDyHead BERT copies.  Returns -1 for shapes it does not take, -3 for misaligned strides.
extern "C" int MQ_SYM(mq_bert_attn_qkv_fwd)(const void* x, const void* w, const void* bias, void* o, const float* key_bias, const int* kv_len,
                                            int B, int T, int C, int H, long x_bs, long x_rs, long o_bs, long o_rs, long bias_bs, float scale,
                                            float clamp, void* stream) {
  if (B <= 0 || T <= 0) return 0;
  if (H <= 0 || C != BA_C || C != BA_D * H || T > 256) return -1;
  if ((x_bs % 8) || (x_rs % 8) || (o_rs % 4) || (o_bs % 4)) return -3;
  BertAttnParams p;
  p.x = (const half_t*)x; p.w = (const half_t*)w; p.bias = (const half_t*)bias; p.o = (half_t*)o; p.key_bias = key_bias; p.kv_len = kv_len;
  p.B = B; p.T = T; p.C = C; p.H = H; p.x_bs = x_bs; p.x_rs = x_rs; p.o_bs = o_bs; p.o_rs = o_rs; p.bias_bs = bias_bs;
  p.scale = scale; p.clamp = clamp;
  p.nblk_cap = (T + 15) / 16;
  hipStream_t s = (hipStream_t)stream;
  const bool small = p.nblk_cap <= 10;
  if (clamp > 0.f) return small ? launch_bert_attn<10, true>(p, s) : launch_bert_attn<16, true>(p, s);
  return small ? launch_bert_attn<10, false>(p, s) : launch_bert_attn<16, false>(p, s);
}

MQ_NAMESPACE_END
