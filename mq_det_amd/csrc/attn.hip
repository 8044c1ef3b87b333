// mq_attn_fwd: fused (flash-style) multi-head attention forward for gfx950.
//
//   O[b, i, h*D:(h+1)*D] = sum_j softmax_j( clamp(scale * Q_i.K_j, +-clamp) + key_bias[b, j] ) * V_j
//
// One kernel family serves every dense attention on the MQ-Det hot path (fp16 in / fp32 accumulate):
//   * BERT self-attention, T=256, 12 heads x 64 (language backbone x12; VLDyHead copy x6 with the
//     +-50000 clamp)                      -- reference rpn/modeling_bert.py:119-170, HF BertSelfAttention
//   * GCP pre-select: vision queries -> pooled image tokens, 8 heads x 32, Nk = 5577
//                                          -- reference modeling_bert_new.py:204-240 (dense branch)
//   * VLFuse image->text (Nq = 22400, Nk = 256, key mask) and text->image (Nq = 256, Nk = 22400,
//     split over keys) attention, 8 heads x 256, +-50000 clamp
//                                          -- reference utils/fuse_helper.py:233-279
// The QK^T logits never reach HBM (the reference materialises them: 183 MB / VLFuse layer / image).
//
// Layout (all "NT", i.e. K-contiguous for the MFMA fragments; see common.h):
//   Q  : element (b,i,h,d) at q  + b*q_bs  + i*q_rs + h*q_hs + d     (q_hs = 0: one Q shared by all heads)
//   K  : element (b,j,h,d) at k  + b*k_bs  + j*k_rs + h*k_hs + d
//   Vt : V transposed, element (b,h,d,j) at vt + b*vt_bs + h*vt_hs + d*vt_rs + j     (strides % 8 == 0)
//   key_bias : fp32 (b,h,j) at key_bias + b*bias_bs + h*bias_hs + j
//   O  : [B, Nq, *]                           o  + b*o_bs  + i*o_rs  + h*D + d
// Work decomposition: grid = (ceil(Nq / BM), B*H, nsplit); a workgroup = 4 waves, each wave owns
// RB*16 query rows and sweeps the keys in tiles of 64 staged through LDS (K tile [64][D+8],
// Vt tile [D][64+8], per-wave P tile).  Online softmax in fp32 with wave-shuffle row reductions.
// nsplit > 1 (few queries, many keys): each split writes un-normalised partials (O, m, l) to a
// workspace and mq_attn_combine merges them.
#include "common.h"

MQ_NAMESPACE_BEGIN

struct AttnParams {
  const half_t* q; const half_t* k; const half_t* vt; half_t* o;
  const float* key_bias;     // [B, Nk] or nullptr
  const int* kv_len;         // [B] or nullptr: keys >= kv_len[b] are masked (contribute exactly 0) -> their tiles are skipped
  const unsigned char* qk_mask;   // per-(query, key) mask, 1 = masked, element (b,h,i,j) at b*mask_bs + h*mask_hs + i*mask_rs + j, or nullptr
  long mask_bs, mask_hs, mask_rs;
  float* ws;                 // split-K workspace or nullptr
  int B, H, Nq, Nk;
  long q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, vt_bs, vt_rs, vt_hs, o_bs, o_rs, bias_bs, bias_hs;
  float scale, clamp;
  int nsplit;
};

template <int D, int RB, bool MASK>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
  constexpr int BM = 4 * RB * 16, BN = 64;
  constexpr int KS = D + 8, VS = BN + 8, PS = BN + 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* Ks = (half_t*)smem;              // [BN][KS]
  half_t* Vs = Ks + BN * KS;               // [D][VS]
  half_t* Ps = Vs + D * VS;                // [4][RB*16][PS]
  constexpr bool QLDS = (D >= 256);        // D=256: Q fragments (64 VGPRs) live in LDS instead, see below
  half_t* Qs = Ps + 4 * RB * 16 * PS;      // [BM][KS] when QLDS
  float* Bias_s = (float*)(Qs + (QLDS ? BM * KS : 0));      // [BN] key bias of the current tile

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  // XCD-aware work order.  Workgroup i runs on XCD i % 8 and every XCD has its own 4 MB L2, so all workgroups that
  // stream the SAME K/V data -- the H heads x q-tiles of one (batch element, key split) "group" -- are placed on ONE XCD,
  // consecutive in time, heads fastest (the heads of one q-tile also share Q when q_hs == 0).  With the naive
  // (q-tile, b*H + h, split) grid the 16 workgroups of a VLFuse text->image group sat on 8 different XCDs and each
  // pulled its 3.8 MB of image tokens through the Infinity-Cache path (~11 B/clk/CU instead of ~50 from L2,
  // profiles/r01_ingest_microbench.txt).
  const int qtiles = (p.Nq + BM - 1) / BM;
  const int members = p.H * qtiles;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int group = (seq / members) * 8 + xcd;                 // group = b * nsplit + split
  if (group >= p.B * p.nsplit) return;
  const int wq = seq % members;
  const int h = wq % p.H, qtile = wq / p.H;
  const int b = group / p.nsplit, split = group % p.nsplit;
  const int bh = b * p.H + h;
  // head strides are free parameters: 0 shares one operand across all heads (folded VLFuse projections)
  const half_t* Q = p.q + (long)b * p.q_bs + (long)h * p.q_hs;
  const half_t* K = p.k + (long)b * p.k_bs + (long)h * p.k_hs;
  const half_t* Vt = p.vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
  const float* bias = p.key_bias ? p.key_bias + (long)b * p.bias_bs + (long)h * p.bias_hs : nullptr;
  half_t* Pw = Ps + wave * (RB * 16 * PS);
  const unsigned char* qmask = MASK ? p.qk_mask + (long)b * p.mask_bs + (long)h * p.mask_hs : nullptr;

  // trailing masked keys (text padding) contribute exactly zero: do not even visit their tiles
  const int nk_eff = p.kv_len ? max(1, min(p.Nk, p.kv_len[b])) : p.Nk;
  const int ntiles = (nk_eff + BN - 1) / BN;
  const int tps = (ntiles + p.nsplit - 1) / p.nsplit;
  const int t0 = split * tps;
  const int t1 = min(ntiles, t0 + tps);
  const int row0 = qtile * BM + wave * (RB * 16);

  half8 qf[QLDS ? 1 : RB][QLDS ? 1 : D / 32];
  if constexpr (QLDS) {
    // the whole register file is needed for O (128 acc regs) + the prefetched K/V tile (64): park Q in LDS
    constexpr int QCH = BM * (D / 8) / 256;
    half8 qtmp[QCH];
#pragma unroll
    for (int i = 0; i < QCH; ++i) {          // all loads in flight first, then the LDS stores
      int c = tid + i * 256;
      int row = min(qtile * BM + c / (D / 8), p.Nq - 1);
      qtmp[i] = *(const half8*)(Q + (long)row * p.q_rs + (c % (D / 8)) * 8);
    }
#pragma unroll
    for (int i = 0; i < QCH; ++i) {
      int c = tid + i * 256;
      *(half8*)(Qs + (c / (D / 8)) * KS + (c % (D / 8)) * 8) = qtmp[i];
    }
  } else {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      int row = min(row0 + rb * 16 + l15, p.Nq - 1);
#pragma unroll
      for (int kk = 0; kk < D / 32; ++kk)
        qf[rb][kk] = *(const half8*)(Q + (long)row * p.q_rs + kk * 32 + lg * 8);
    }
  }
  const half_t* Qw = Qs + (wave * RB * 16) * KS;
  float4_ o[RB][D / 16];
  float m[RB][4], lsum[RB][4];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
    for (int db = 0; db < D / 16; ++db) o[rb][db] = (float4_){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[rb][r] = MQ_NEG_BIG; lsum[rb][r] = 0.f; }
  }

  // ---- software pipeline: the global loads of tile t+1 are issued before the math of tile t and land in
  // registers while the MFMAs run; they are committed to LDS at the top of the next iteration.  (v1 issued
  // one load -> LDS store at a time: 16 exposed L2 round trips per tile, ~28k cycles vs ~2.5k of MFMA work.)
  // Out-of-range keys are NOT zero-filled: addresses are clamped to valid data and the logits of those
  // columns are forced to -1e30 below (P == 0 exactly), so they contribute nothing.
  constexpr int NCH = (BN * (D / 8)) / 256;          // 16-byte chunks per thread per tile (K and V each)
  static_assert((BN * (D / 8)) % 256 == 0, "tile chunks must divide the block");
  half8 kreg[NCH], vreg[NCH];
  float breg = 0.f;                                    // key-bias element of the prefetched tile (threads < BN)
  const int vlast = ((p.Nk - 1) / 8) * 8;            // last valid 8-key chunk start of a Vt row
  auto issue = [&](int t) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int c = tid + i * 256;
      int r = c / (D / 8), ch = c % (D / 8);
      int key = min(t * BN + r, p.Nk - 1);
      kreg[i] = *(const half8*)(K + (long)key * p.k_rs + ch * 8);
      int d = c / (BN / 8), cv = c % (BN / 8);
      int key0 = min(t * BN + cv * 8, vlast);
      vreg[i] = *(const half8*)(Vt + (long)d * p.vt_rs + key0);
    }
    // the bias rides with the tile prefetch: any OTHER load consumed inside the iteration makes hipcc drain the whole
    // VMEM queue (s_waitcnt vmcnt(0)) right after the prefetch is issued, which serialises it (seen in the ISA of v2)
    if (bias && tid < BN) breg = bias[min(t * BN + tid, p.Nk - 1)];
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int c = tid + i * 256;
      *(half8*)(Ks + (c / (D / 8)) * KS + (c % (D / 8)) * 8) = kreg[i];
      *(half8*)(Vs + (c / (BN / 8)) * VS + (c % (BN / 8)) * 8) = vreg[i];
    }
    if (tid < BN) Bias_s[tid] = breg;
  };
  if (t0 < t1) issue(t0);

  constexpr float THR = 8.0f;     // deferred rescale: O / l are rescaled only when a row max grows by > THR

  for (int t = t0; t < t1; ++t) {
    __syncthreads();                                  // every wave is done reading the previous tiles
    commit();
    __syncthreads();
    float kbias[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) kbias[nb] = Bias_s[nb * 16 + l15];
    if (t + 1 < t1) issue(t + 1);

    // ---- S = Q K^T  (RB x 4 blocks of 16x16 per wave).  LDS fragment reads are software-pipelined one
    // k-step ahead of the MFMAs that consume them: with one wave per SIMD nothing else hides LDS latency.
    float4_ s[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) s[rb][nb] = (float4_){0.f, 0.f, 0.f, 0.f};
    {
      half8 qa[2][RB], kf[2][4];
      auto ld = [&](int kk, int buf) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          if constexpr (QLDS) qa[buf][rb] = *(const half8*)(Qw + (rb * 16 + l15) * KS + kk * 32 + lg * 8);
          else qa[buf][rb] = qf[rb][kk];
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) kf[buf][nb] = *(const half8*)(Ks + (nb * 16 + l15) * KS + kk * 32 + lg * 8);
      };
      ld(0, 0);
#pragma unroll
      for (int kk = 0; kk < D / 32; ++kk) {
        const int cur = kk & 1;
        if (kk + 1 < D / 32) ld(kk + 1, cur ^ 1);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) s[rb][nb] = mfma16(qa[cur][rb], kf[cur][nb], s[rb][nb]);
      }
    }
    // ---- scale / key bias / clamp / masks
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      int key = t * BN + nb * 16 + l15;
      bool valid = key < p.Nk;
      // key_bias carries a finite additive term (folded projection bias, added BEFORE the clamp like the
      // reference's q.k logits) and/or the padding mask (<= -1e29 => the key is masked AFTER the clamp).
      float kb = kbias[nb];
      const bool masked = kb < -1.0e29f;
      if (masked) kb = 0.f;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = s[rb][nb][r] * p.scale + kb;
          if (p.clamp > 0.f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
          bool on = valid && !masked;
          if constexpr (MASK) {                            // sub-sentence / attn_mask style masks (text-sized attentions only)
            const int row = min(row0 + rb * 16 + lg * 4 + r, p.Nq - 1);
            on = on && qmask[(long)row * p.mask_rs + min(key, p.Nk - 1)] == 0;
          }
          s[rb][nb][r] = on ? v : MQ_NEG_BIG;
        }
    }
    // ---- online softmax with deferred rescale: the decision is taken BEFORE this tile's P is exponentiated and
    // after the previous tile's P.V has been fully accumulated, so O, l and P always share one reference max.
    float mx[RB][4];
    bool grow = false;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = fmaxf(fmaxf(s[rb][0][r], s[rb][1][r]), fmaxf(s[rb][2][r], s[rb][3][r]));
        mx[rb][r] = group16_max(v);
        grow |= mx[rb][r] > m[rb][r] + THR;
      }
    if (__any(grow)) {                                  // wave-uniform, rare after the first tiles
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float mnew = fmaxf(m[rb][r], mx[rb][r]);
          float alpha = __expf(m[rb][r] - mnew);
          lsum[rb][r] *= alpha;
          m[rb][r] = mnew;
#pragma unroll
          for (int db = 0; db < D / 16; ++db) o[rb][db][r] *= alpha;
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float rs = 0.f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          float pv = __expf(s[rb][nb][r] - m[rb][r]);      // <= e^THR, exact softmax after the final 1/l
          rs += pv;
          Pw[(rb * 16 + lg * 4 + r) * PS + nb * 16 + l15] = (half_t)pv;
        }
        lsum[rb][r] += group16_sum(rs);
      }
    wave_lds_fence();
    // ---- O += P V   (A = P from LDS, B = V from the transposed tile), V fragments prefetched 4 blocks ahead
#pragma unroll
    for (int kk = 0; kk < BN / 32; ++kk) {
      half8 pf[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) pf[rb] = *(const half8*)(Pw + (rb * 16 + l15) * PS + kk * 32 + lg * 8);
      constexpr int G = (D / 16) < 4 ? (D / 16) : 4, NG = (D / 16) / G;
      half8 vf[2][G];
      auto ldv = [&](int g, int buf) {
#pragma unroll
        for (int i = 0; i < G; ++i) vf[buf][i] = *(const half8*)(Vs + ((g * G + i) * 16 + l15) * VS + kk * 32 + lg * 8);
      };
      ldv(0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int cur = g & 1;
        if (g + 1 < NG) ldv(g + 1, cur ^ 1);
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) o[rb][g * G + i] = mfma16(pf[rb], vf[cur][i], o[rb][g * G + i]);
      }
    }
  }

  // ---- epilogue
  if (p.nsplit == 1) {
    // transpose through LDS (the K/V tiles are dead) so that every row leaves as 16-byte coalesced stores
    constexpr int OS = D + 8;
    static_assert(4 * RB * 16 * OS <= BN * KS + D * VS + 4 * RB * 16 * PS + (QLDS ? 0 : 0), "O staging must fit in the LDS tiles");
    __syncthreads();
    half_t* Ow = Ks + wave * (RB * 16 * OS);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float inv = 1.f / lsum[rb][r];
#pragma unroll
        for (int db = 0; db < D / 16; ++db)
          Ow[(rb * 16 + lg * 4 + r) * OS + db * 16 + l15] = (half_t)(o[rb][db][r] * inv);
      }
    wave_lds_fence();
    half_t* O = p.o + (long)b * p.o_bs + h * D;
    for (int c = lane; c < RB * 16 * (D / 8); c += 64) {
      int rr = c / (D / 8), ch = c % (D / 8);
      int row = row0 + rr;
      if (row < p.Nq) *(half8*)(O + (long)row * p.o_rs + ch * 8) = *(const half8*)(Ow + rr * OS + ch * 8);
    }
  } else {
    // workspace: [nsplit][B*H][Nq][D + 2] floats  (O unnormalised, then m, l)
    float* W = p.ws + ((long)split * (p.B * p.H) + bh) * (long)p.Nq * (D + 2);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + rb * 16 + lg * 4 + r;
        if (row < p.Nq) {
          float* wr = W + (long)row * (D + 2);
#pragma unroll
          for (int db = 0; db < D / 16; ++db) wr[db * 16 + l15] = o[rb][db][r];
          if (l15 == 0) { wr[D] = m[rb][r]; wr[D + 1] = lsum[rb][r]; }
        }
      }
  }
}

// merge split-K partials: one wave per (b*h, row)
template <int D>
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long total = (long)p.B * p.H * p.Nq;
  if (gw >= total) return;
  const int row = gw % p.Nq;
  const int bh = gw / p.Nq, b = bh / p.H, h = bh % p.H;
  const long stride = (long)p.B * p.H * p.Nq * (D + 2);
  const float* base = p.ws + gw * (D + 2);
  float mx = MQ_NEG_BIG;
  for (int s = 0; s < p.nsplit; ++s) mx = fmaxf(mx, base[s * stride + D]);
  float l = 0.f;
  constexpr int NA = D >= 64 ? D / 64 : 1;
  float acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = 0.f;
  float acc_small = 0.f;     // D < 64 (D = 32): lanes 0..31 only
  for (int s = 0; s < p.nsplit; ++s) {
    const float* w = base + s * stride;
    float f = __expf(w[D] - mx);
    l += w[D + 1] * f;
    if (D >= 64) {
#pragma unroll
      for (int i = 0; i < D / 64; ++i) acc[i] += w[i * 64 + lane] * f;
    } else if (lane < D) {
      acc_small += w[lane] * f;
    }
  }
  half_t* O = p.o + (long)b * p.o_bs + (long)row * p.o_rs + h * D;
  float inv = 1.f / l;
  if (D >= 64) {
#pragma unroll
    for (int i = 0; i < D / 64; ++i) O[i * 64 + lane] = (half_t)(acc[i] * inv);
  } else if (lane < D) {
    O[lane] = (half_t)(acc_small * inv);
  }
}

template <int D, int RB, bool MASK>
static int launch_attn(const AttnParams& p, hipStream_t stream) {
  constexpr int BM = 4 * RB * 16, BN = 64;
  constexpr size_t smem = (size_t)(BN * (D + 8) + D * (BN + 8) + 4 * RB * 16 * (BN + 8) + (D >= 256 ? BM * (D + 8) : 0)) * sizeof(half_t) + BN * sizeof(float);
  static MqOncePerDevice attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel<D, RB, MASK>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set.done();
  }
  const int groups = p.B * p.nsplit, members = p.H * ((p.Nq + BM - 1) / BM);
  dim3 grid((unsigned)(8 * ((groups + 7) / 8) * members));
  hipLaunchKernelGGL((attn_fwd_kernel<D, RB, MASK>), grid, dim3(256), smem, stream, p);
  MQ_CHECK_LAUNCH();
  if (p.nsplit > 1) {
    long total = (long)p.B * p.H * p.Nq;
    hipLaunchKernelGGL((attn_combine_kernel<D>), dim3((unsigned)((total + 3) / 4)), dim3(256), 0, stream, p);
    MQ_CHECK_LAUNCH();
  }
  return 0;
}

#ifdef MQ_PRIMARY_UNIT
extern "C" long mq_attn_workspace_bytes(int B, int H, int Nq, int D, int nsplit) {
  return nsplit > 1 ? (long)nsplit * B * H * Nq * (D + 2) * (long)sizeof(float) : 0;
}
#endif

extern "C" int MQ_SYM(mq_attn_fwd)(const void* q, const void* k, const void* vt, void* o, const float* key_bias,
                           const int* kv_len, const unsigned char* qk_mask, long mask_bs, long mask_hs, long mask_rs,
                           void* workspace, int B, int H, int Nq, int Nk, int D,
                           long q_bs, long q_rs, long q_hs, long k_bs, long k_rs, long k_hs,
                           long vt_bs, long vt_rs, long vt_hs, long o_bs, long o_rs, long bias_bs, long bias_hs,
                           float scale, float clamp, int nsplit, void* stream) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return 0;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > 1 && workspace == nullptr) return -2;
  if ((vt_rs % 8) || (q_rs % 8) || (k_rs % 8) || (q_hs % 8) || (k_hs % 8) || (vt_hs % 8)) return -3;
  AttnParams p;
  p.q = (const half_t*)q; p.k = (const half_t*)k; p.vt = (const half_t*)vt; p.o = (half_t*)o;
  p.key_bias = key_bias; p.kv_len = kv_len; p.ws = (float*)workspace;
  p.qk_mask = qk_mask; p.mask_bs = mask_bs; p.mask_hs = mask_hs; p.mask_rs = mask_rs;
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
  p.q_bs = q_bs; p.q_rs = q_rs; p.q_hs = q_hs; p.k_bs = k_bs; p.k_rs = k_rs; p.k_hs = k_hs;
  p.vt_bs = vt_bs; p.vt_rs = vt_rs; p.vt_hs = vt_hs; p.bias_bs = bias_bs; p.bias_hs = bias_hs;
  p.o_bs = o_bs; p.o_rs = o_rs; p.scale = scale; p.clamp = clamp; p.nsplit = nsplit;
  hipStream_t s = (hipStream_t)stream;
  if (qk_mask) {
    switch (D) {
      case 32: return launch_attn<32, 2, true>(p, s);
      case 64: return launch_attn<64, 2, true>(p, s);
      default: return -1;
    }
  }
  switch (D) {
    case 32: return launch_attn<32, 2, false>(p, s);
    case 64: return launch_attn<64, 2, false>(p, s);
    default: return -1;
  }
}

MQ_NAMESPACE_END
