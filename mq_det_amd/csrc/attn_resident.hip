// mq_attn_resident_fwd: multi-head attention for SHORT key sequences (Nk <= 256: every attention over the text tokens -- BERT self-
// attention 12 x 64 over T = 256, the VLDyHead copy with the +-50000 clamp, MQ-GroundingDINO's text enhancer 4 x 64 and decoder
// text cross-attention 8 x 32), gfx950.  Same operator and operand layout as mq_attn_fwd (attn.hip); a different schedule:
//
//   * all keys of one (batch, head) are RESIDENT in LDS: K [Nk][D] and V^T [D][Nk] are staged once per workgroup (72 KB for D = 64),
//     one barrier per workgroup instead of two per 64-key tile;
//   * S^T = K Q^T is computed instead of S (operands swapped, like vlfuse_attn.hip): the lane that owns query column `l & 15` of an
//     S^T block holds keys 4 (l >> 4) .. + 3 of it, which is exactly half of the B fragment (k-slots 8 (l >> 4) ..) the P V product
//     wants -- two 16-key blocks make one 32-key k-step, the matching A fragment is two 8-byte reads of a V^T row.  P never goes
//     through LDS (mq_attn_fwd: 32 two-byte LDS stores + a fence + 4 b128 reads per lane and tile), row reductions are within the
//     lane plus 2 shuffles (there: 4), the 1 / l normalisation is a per-lane scalar;
//   * all <= 256 logits of a query stay in registers: exact two-pass softmax, no running max / rescale;
//   * exp2 domain: log2(e) is folded into the scale and the bias, one v_exp_f32 per logit without the multiply.
// The host routes text-sized calls here (KERNELS["ATTN_RESIDENT"] = 1, the default since round 3: +4.7 % end to end on its own,
// profiles/r03_call1_switch_ab.txt; B = 64 language path: attention launches 5.0 -> 9.4 % MFMA); mq_attn_fwd stays the kernel for
// per-(query, key) masks and is selectable for everything.
//
// Work decomposition: grid = ceil(Nq / 128) x B x H workgroups of 4 waves, a wave owns 32 queries (2 column blocks); 2 workgroups
// fit a CU (LDS 2 x 72 KB, <= 256 VGPRs), so one workgroup's K / V fill overlaps the other's MFMAs.
#include "common.h"

MQ_NAMESPACE_BEGIN

struct ResAttnParams {
  const half_t* q; const half_t* k; const half_t* vt; half_t* o;
  const float* key_bias;          // (b, h, j) at key_bias + b*bias_bs + h*bias_hs + j, or nullptr
  const int* kv_len;              // [B] or nullptr: 16-key blocks at and beyond kv_len[b] are skipped (the caller's key_bias masks the rest)
  const unsigned char* qk_mask;   // per-(query, key) byte mask or nullptr
  long mask_bs, mask_hs, mask_rs;
  int B, H, Nq, Nk;
  long q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, vt_bs, vt_rs, vt_hs, o_bs, o_rs, bias_bs, bias_hs;
  float scale, clamp;
};

namespace {
constexpr int RES_NKMAX = 256, RES_BM = 128, RES_QB = 2;
constexpr float RES_LOG2E = 1.4426950408889634f;
}

// CLAMP (the +-50000 clamp of the VLDyHead BERT copies): logit = med3(s * scale + bias, +-clamp) + key mask -- three VALU operations;
// without it the mask rides in the bias term of the one fma (-1e30 absorbs any finite logit).
template <int D, bool MASK, bool CLAMP>
__global__ __launch_bounds__(256) void attn_resident_kernel(ResAttnParams p) {
  constexpr int QB = RES_QB, KS = D + 8, VS = RES_NKMAX + 8, NB = RES_NKMAX / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* Ks = (half_t*)smem;                       // [256][KS]   key rows
  half_t* Vs = Ks + RES_NKMAX * KS;                 // [D][VS]     V^T rows (pitch 528 B: the b64 fragment reads are conflict-free)
  float* Bias_s = (float*)(Vs + D * VS);            // [256]       log2(e) x key bias; masked / out-of-range keys: -1e30 (CLAMP: 0)
  float* Kmask_s = Bias_s + RES_NKMAX;              // [256]       CLAMP only: 0, or -1e30 for masked / out-of-range keys

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int qtiles = (p.Nq + RES_BM - 1) / RES_BM;
  // consecutive workgroups = the q-tiles and heads of one batch element (they share K / V / bias lines in the XCD's L2 only by
  // luck of the i % 8 XCD assignment -- the operands are small: 64 KB per (b, h))
  const int qtile = blockIdx.x % qtiles;
  const int bh = blockIdx.x / qtiles, h = bh % p.H, b = bh / p.H;
  const half_t* Q = p.q + (long)b * p.q_bs + (long)h * p.q_hs;
  const half_t* K = p.k + (long)b * p.k_bs + (long)h * p.k_hs;
  const half_t* Vt = p.vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
  const int nk_eff = p.kv_len ? max(1, min(p.Nk, p.kv_len[b])) : p.Nk;
  const int nblk = (nk_eff + 15) >> 4;              // 16-key blocks visited (wave-uniform)
  const int nst = (nblk + 1) >> 1;                  // 32-key steps of the P V product

  // ---- stage K rows [0, 16 nblk), V^T columns [0, 32 nst) and the bias; out-of-range addresses are clamped to valid (finite) data,
  // their logits are forced to -1e30 below, so they contribute exactly 0
  {
    // all global loads of a thread are issued before the first LDS store (one memory round trip, not one per chunk)
    constexpr int KCH = RES_NKMAX * (D / 8) / 256, VCH = D * (RES_NKMAX / 8) / 256;     // 16-byte chunks per thread: 8 + 8 (D = 64)
    const int kchunks = nblk * 16 * (D / 8), vcols8 = nst * 4;                         // 8-key chunks per V^T row
    const int vlast = ((p.Nk - 1) / 8) * 8;
    half8 kreg[KCH], vreg[VCH];
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int c = tid + i * 256, r = c / (D / 8), ch = c % (D / 8);
      if (c < kchunks) kreg[i] = *(const half8*)(K + (long)min(r, p.Nk - 1) * p.k_rs + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
      const int c = tid + i * 256, d = c >> 5, cv = c & 31;
      if (cv < vcols8) vreg[i] = *(const half8*)(Vt + (long)d * p.vt_rs + min(cv * 8, vlast));
    }
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int c = tid + i * 256, r = c / (D / 8), ch = c % (D / 8);
      if (c < kchunks) *(half8*)(Ks + r * KS + ch * 8) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
      const int c = tid + i * 256, d = c >> 5, cv = c & 31;
      if (cv < vcols8) *(half8*)(Vs + d * VS + cv * 8) = vreg[i];
    }
    for (int j = tid; j < nst * 32; j += 256) {
      float kb = MQ_NEG_BIG;
      if (j < p.Nk) kb = p.key_bias ? p.key_bias[(long)b * p.bias_bs + (long)h * p.bias_hs + j] : 0.f;
      const bool masked = kb < -1.0e29f;                          // <= -1e29: the key is masked (AFTER the clamp, like mq_attn_fwd)
      if constexpr (CLAMP) {
        Bias_s[j] = masked ? 0.f : kb * RES_LOG2E;
        Kmask_s[j] = masked ? MQ_NEG_BIG : 0.f;
      } else {
        Bias_s[j] = masked ? MQ_NEG_BIG : kb * RES_LOG2E;
      }
    }
  }
  const int row0 = qtile * RES_BM + wave * (QB * 16);
  half8 qf[QB][D / 32];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int row = min(row0 + qb * 16 + l15, p.Nq - 1);
#pragma unroll
    for (int kk = 0; kk < D / 32; ++kk) qf[qb][kk] = *(const half8*)(Q + (long)row * p.q_rs + kk * 32 + lg * 8);
  }
  // the per-(query, key) mask: 4 keys = one 32-bit word per (block, query column), all words in flight before the barrier
  // (the host guarantees Nk % 4 == 0, mask strides % 4 == 0 and a 4-byte aligned base, mq_attn_resident_fwd)
  unsigned mw[MASK ? RES_NKMAX / 16 : 1][QB];
  if constexpr (MASK) {
    const unsigned char* qmask = p.qk_mask + (long)b * p.mask_bs + (long)h * p.mask_hs;
#pragma unroll
    for (int nb = 0; nb < RES_NKMAX / 16; ++nb)
      if (nb < nblk) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const int row = min(row0 + qb * 16 + l15, p.Nq - 1);
          mw[nb][qb] = *(const unsigned*)(qmask + (long)row * p.mask_rs + min(nb * 16 + lg * 4, p.Nk - 4));
        }
      }
  }
  __syncthreads();
  if (row0 >= p.Nq) return;                          // whole wave beyond the last query (no barrier after this point)

  // ---- S^T[nb][qb] = K_block . Q^T, then scale / bias / clamp / masks in the log2 domain
  float4_ s[NB][QB];
  const float sc2 = p.scale * RES_LOG2E, cl2 = p.clamp * RES_LOG2E;
  float mx[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) mx[qb] = MQ_NEG_BIG;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if (nb < nblk) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) s[nb][qb] = (float4_){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < D / 32; ++kk) {
        const half8 kf = *(const half8*)(Ks + (nb * 16 + l15) * KS + kk * 32 + lg * 8);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s[nb][qb] = mfma16(kf, qf[qb][kk], s[nb][qb]);
      }
      const float4_ kb4 = *(const float4_*)(Bias_s + nb * 16 + lg * 4);
      float4_ km4 = (float4_){0.f, 0.f, 0.f, 0.f};
      if constexpr (CLAMP) km4 = *(const float4_*)(Kmask_s + nb * 16 + lg * 4);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = __builtin_fmaf(s[nb][qb][r], sc2, kb4[r]);
          if constexpr (CLAMP) v = __builtin_amdgcn_fmed3f(v, -cl2, cl2) + km4[r];
          if constexpr (MASK) v = ((mw[nb][qb] >> (8 * r)) & 0xFFu) ? MQ_NEG_BIG : v;      // keys >= Nk: already -1e30 through the bias
          s[nb][qb][r] = v;
          mx[qb] = fmaxf(mx[qb], v);
        }
      }
    }
  }
  // ---- exact softmax: row max and sum over all keys of a query = over this lane's 4 nblk values and the 4 lanes l15 + 16 * {0..3}
  float lsum[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    mx[qb] = fmaxf(mx[qb], __shfl_xor(mx[qb], 16));
    mx[qb] = fmaxf(mx[qb], __shfl_xor(mx[qb], 32));
    lsum[qb] = 0.f;
  }
  // ---- O^T[db][qb] = V^T . P^T, 32 keys per step: blocks 2 st (k-slots 0..3 of a lane) and 2 st + 1 (k-slots 4..7)
  float4_ o[D / 16][QB];
#pragma unroll
  for (int db = 0; db < D / 16; ++db)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) o[db][qb] = (float4_){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int st = 0; st < NB / 2; ++st) {
    if (st < nst) {
      half8 pf[QB];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p0 = __builtin_amdgcn_exp2f(s[2 * st][qb][r] - mx[qb]);
          const float p1 = (2 * st + 1 < nblk) ? __builtin_amdgcn_exp2f(s[2 * st + 1][qb][r] - mx[qb]) : 0.f;
          lsum[qb] += p0 + p1;
          pf[qb][r] = (half_t)p0;
          pf[qb][4 + r] = (half_t)p1;
        }
      }
#pragma unroll
      for (int db = 0; db < D / 16; ++db) {
        const half_t* vrow = Vs + (db * 16 + l15) * VS + st * 32 + lg * 4;
        const half4 lo = *(const half4*)vrow;
        const half4 hi = *(const half4*)(vrow + 16);
        half8 a;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = lo[j]; a[4 + j] = hi[j]; }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) o[db][qb] = mfma16(a, pf[qb], o[db][qb]);
      }
    }
  }
  // ---- epilogue: 1 / l per query (per lane), 4 consecutive channels of one query per lane -> 8-byte stores
  half_t* O = p.o + (long)b * p.o_bs + h * D;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float l = lsum[qb];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    const int row = row0 + qb * 16 + l15;
    if (row < p.Nq) {
#pragma unroll
      for (int db = 0; db < D / 16; ++db) {
        half4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (half_t)(o[db][qb][r] * inv);
        *(half4*)(O + (long)row * p.o_rs + db * 16 + lg * 4) = v;
      }
    }
  }
}

template <int D, bool MASK, bool CLAMP>
static int launch_resident_c(const ResAttnParams& p, hipStream_t stream) {
  constexpr size_t smem = (size_t)(RES_NKMAX * (D + 8) + D * (RES_NKMAX + 8)) * sizeof(half_t) + 2 * RES_NKMAX * sizeof(float);
  static MqOncePerDevice attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_resident_kernel<D, MASK, CLAMP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set.done();
  }
  const int qtiles = (p.Nq + RES_BM - 1) / RES_BM;
  hipLaunchKernelGGL((attn_resident_kernel<D, MASK, CLAMP>), dim3((unsigned)(qtiles * p.B * p.H)), dim3(256), smem, stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

template <int D, bool MASK>
static int launch_resident(const ResAttnParams& p, hipStream_t stream) {
  return p.clamp > 0.f ? launch_resident_c<D, MASK, true>(p, stream) : launch_resident_c<D, MASK, false>(p, stream);
}

// Same arguments as mq_attn_fwd without the key split (Nk <= 256, D in {32, 64}); see include/mqdet_hip.h.
extern "C" int MQ_SYM(mq_attn_resident_fwd)(const void* q, const void* k, const void* vt, void* o, const float* key_bias,
                                            const int* kv_len, const unsigned char* qk_mask, long mask_bs, long mask_hs, long mask_rs,
                                            int B, int H, int Nq, int Nk, int D,
                                            long q_bs, long q_rs, long q_hs, long k_bs, long k_rs, long k_hs,
                                            long vt_bs, long vt_rs, long vt_hs, long o_bs, long o_rs, long bias_bs, long bias_hs,
                                            float scale, float clamp, void* stream) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return 0;
  if (Nk > RES_NKMAX) return -1;
  if ((vt_rs % 8) || (q_rs % 8) || (k_rs % 8) || (q_hs % 8) || (k_hs % 8) || (vt_hs % 8) || (o_rs % 4)) return -3;
  if (qk_mask && ((Nk % 4) || (mask_bs % 4) || (mask_hs % 4) || (mask_rs % 4) || ((uintptr_t)qk_mask % 4))) return -3;
  ResAttnParams p;
  p.q = (const half_t*)q; p.k = (const half_t*)k; p.vt = (const half_t*)vt; p.o = (half_t*)o;
  p.key_bias = key_bias; p.kv_len = kv_len;
  p.qk_mask = qk_mask; p.mask_bs = mask_bs; p.mask_hs = mask_hs; p.mask_rs = mask_rs;
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
  p.q_bs = q_bs; p.q_rs = q_rs; p.q_hs = q_hs; p.k_bs = k_bs; p.k_rs = k_rs; p.k_hs = k_hs;
  p.vt_bs = vt_bs; p.vt_rs = vt_rs; p.vt_hs = vt_hs; p.bias_bs = bias_bs; p.bias_hs = bias_hs;
  p.o_bs = o_bs; p.o_rs = o_rs; p.scale = scale; p.clamp = clamp;
  hipStream_t s = (hipStream_t)stream;
  if (qk_mask) {
    switch (D) {
      case 32: return launch_resident<32, true>(p, s);
      case 64: return launch_resident<64, true>(p, s);
      default: return -1;
    }
  }
  switch (D) {
    case 32: return launch_resident<32, false>(p, s);
    case 64: return launch_resident<64, false>(p, s);
    default: return -1;
  }
}


// ------------------------------------------------------------------------------------------------------------------------------
// mq_attn_chunked_fwd: the same S^T formulation for LONG key sequences (GCP pre-select: 200 vision queries x 5577 pooled image tokens,
// 8 x 32; MQ-GroundingDINO decoder self-attention 900 x 900).  Keys are processed in chunks of 256: within a chunk everything is the
// resident kernel (all 256 logits of a query in registers, exact max), between chunks a running (max, sum, O) is rescaled ONCE per
// 256 keys -- mq_attn_fwd decides about a rescale every 64 keys and moves P through LDS.  The next chunk's K / V^T travel in a
// register prefetch ring while the current one is on the MFMAs and are committed to the other LDS buffer: one barrier per chunk.
// Key split (nsplit > 1) writes the same (O, m, l) partials as mq_attn_fwd; a combine kernel merges them.
// Selected like the resident kernel (KERNELS["ATTN_RESIDENT"] = 1, default).
template <int D, bool CLAMP>
__global__ __launch_bounds__(256) void attn_chunked_kernel(ResAttnParams p, float* ws, int nsplit) {
  constexpr int QB = RES_QB, KS = D + 8, VS = RES_NKMAX + 8, NB = RES_NKMAX / 16, CH = RES_NKMAX;
  constexpr int BUF_HALFS = CH * KS + D * VS;                       // one buffer: K chunk + V^T chunk
  constexpr int KCH = CH * (D / 8) / 256, VCH = D * (CH / 8) / 256;  // 16-byte chunks per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* Tiles = (half_t*)smem;                                    // [2][BUF_HALFS]
  float* Bias_all = (float*)(Tiles + 2 * BUF_HALFS);                // [2][2][256]: per buffer (bias, key mask)

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int qtiles = (p.Nq + RES_BM - 1) / RES_BM;
  int idx = blockIdx.x;
  const int qtile = idx % qtiles; idx /= qtiles;
  const int h = idx % p.H; idx /= p.H;
  const int split = idx % nsplit, b = idx / nsplit;
  const half_t* Q = p.q + (long)b * p.q_bs + (long)h * p.q_hs;
  const half_t* K = p.k + (long)b * p.k_bs + (long)h * p.k_hs;
  const half_t* Vt = p.vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
  const int nk_eff = p.kv_len ? max(1, min(p.Nk, p.kv_len[b])) : p.Nk;
  const int nchunks = (nk_eff + CH - 1) / CH;
  const int cps = (nchunks + nsplit - 1) / nsplit;
  const int c0 = split * cps, c1 = min(nchunks, c0 + cps);
  const int vlast = ((p.Nk - 1) / 8) * 8;

  half8 kreg[KCH], vreg[VCH];
  float breg = 0.f;                                                  // bias element tid of the prefetched chunk
  auto issue = [&](int c) {                                          // global loads of chunk c (addresses clamped to valid data)
    const int key0 = c * CH;
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int cc = tid + i * 256, r = cc / (D / 8), ch = cc % (D / 8);
      kreg[i] = *(const half8*)(K + (long)min(key0 + r, p.Nk - 1) * p.k_rs + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
      const int cc = tid + i * 256, d = cc >> 5, cv = cc & 31;
      vreg[i] = *(const half8*)(Vt + (long)d * p.vt_rs + min(key0 + cv * 8, vlast));
    }
    const int j = key0 + tid;
    breg = MQ_NEG_BIG;
    if (j < nk_eff) breg = p.key_bias ? p.key_bias[(long)b * p.bias_bs + (long)h * p.bias_hs + j] : 0.f;
  };
  auto commit = [&](int buf) {
    half_t* Ks = Tiles + buf * BUF_HALFS;
    half_t* Vs = Ks + CH * KS;
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int cc = tid + i * 256, r = cc / (D / 8), ch = cc % (D / 8);
      *(half8*)(Ks + r * KS + ch * 8) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
      const int cc = tid + i * 256, d = cc >> 5, cv = cc & 31;
      *(half8*)(Vs + d * VS + cv * 8) = vreg[i];
    }
    const bool masked = breg < -1.0e29f;
    float* Bs = Bias_all + buf * 2 * CH;
    if constexpr (CLAMP) {
      Bs[tid] = masked ? 0.f : breg * RES_LOG2E;
      Bs[CH + tid] = masked ? MQ_NEG_BIG : 0.f;
    } else {
      Bs[tid] = masked ? MQ_NEG_BIG : breg * RES_LOG2E;
    }
  };

  const int row0 = qtile * RES_BM + wave * (QB * 16);
  half8 qf[QB][D / 32];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int row = min(row0 + qb * 16 + l15, p.Nq - 1);
#pragma unroll
    for (int kk = 0; kk < D / 32; ++kk) qf[qb][kk] = *(const half8*)(Q + (long)row * p.q_rs + kk * 32 + lg * 8);
  }
  float4_ o[D / 16][QB];
  float m_run[QB], l_run[QB];                                        // running max (log2 domain, all 4 lanes of a query agree) and this lane's partial sum
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = MQ_NEG_BIG;
    l_run[qb] = 0.f;
#pragma unroll
    for (int db = 0; db < D / 16; ++db) o[db][qb] = (float4_){0.f, 0.f, 0.f, 0.f};
  }
  const float sc2 = p.scale * RES_LOG2E, cl2 = p.clamp * RES_LOG2E;
  if (c0 < c1) issue(c0);

  for (int c = c0; c < c1; ++c) {
    const int buf = (c - c0) & 1;
    commit(buf);                                                     // buffer `buf` was last read two chunks ago (barrier of chunk c - 1 passed)
    __syncthreads();
    if (c + 1 < c1) issue(c + 1);
    const half_t* Ks = Tiles + buf * BUF_HALFS;
    const half_t* Vs = Ks + CH * KS;
    const float* Bs = Bias_all + buf * 2 * CH;
    const int nblk = min(NB, (nk_eff - c * CH + 15) >> 4), nst = (nblk + 1) >> 1;

    float4_ s[NB][QB];
    float mx[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) mx[qb] = MQ_NEG_BIG;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (nb < nblk) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s[nb][qb] = (float4_){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < D / 32; ++kk) {
          const half8 kf = *(const half8*)(Ks + (nb * 16 + l15) * KS + kk * 32 + lg * 8);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) s[nb][qb] = mfma16(kf, qf[qb][kk], s[nb][qb]);
        }
        const float4_ kb4 = *(const float4_*)(Bs + nb * 16 + lg * 4);
        float4_ km4 = (float4_){0.f, 0.f, 0.f, 0.f};
        if constexpr (CLAMP) km4 = *(const float4_*)(Bs + CH + nb * 16 + lg * 4);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = __builtin_fmaf(s[nb][qb][r], sc2, kb4[r]);
            if constexpr (CLAMP) v = __builtin_amdgcn_fmed3f(v, -cl2, cl2) + km4[r];
            s[nb][qb][r] = v;
            mx[qb] = fmaxf(mx[qb], v);
          }
      }
    }
    // running max over the chunks; O and the partial sums move to the new reference once per chunk
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      mx[qb] = fmaxf(mx[qb], __shfl_xor(mx[qb], 16));
      mx[qb] = fmaxf(mx[qb], __shfl_xor(mx[qb], 32));
      const float m_new = fmaxf(m_run[qb], mx[qb]);
      const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);  // first chunk: exp2(-1e30 - m) = 0 (O and l are 0 anyway)
      m_run[qb] = m_new;
      l_run[qb] *= alpha;
#pragma unroll
      for (int db = 0; db < D / 16; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[db][qb][r] *= alpha;
    }
#pragma unroll
    for (int st = 0; st < NB / 2; ++st) {
      if (st < nst) {
        half8 pf[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p0 = __builtin_amdgcn_exp2f(s[2 * st][qb][r] - m_run[qb]);
            const float p1 = (2 * st + 1 < nblk) ? __builtin_amdgcn_exp2f(s[2 * st + 1][qb][r] - m_run[qb]) : 0.f;
            l_run[qb] += p0 + p1;
            pf[qb][r] = (half_t)p0;
            pf[qb][4 + r] = (half_t)p1;
          }
        }
#pragma unroll
        for (int db = 0; db < D / 16; ++db) {
          const half_t* vrow = Vs + (db * 16 + l15) * VS + st * 32 + lg * 4;
          const half4 lo = *(const half4*)vrow;
          const half4 hi = *(const half4*)(vrow + 16);
          half8 a;
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[j] = lo[j]; a[4 + j] = hi[j]; }
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) o[db][qb] = mfma16(a, pf[qb], o[db][qb]);
        }
      }
    }
  }

  // ---- epilogue
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float l = l_run[qb];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const int row = row0 + qb * 16 + l15;
    if (row < p.Nq) {
      if (nsplit == 1) {
        const float inv = 1.f / l;
        half_t* O = p.o + (long)b * p.o_bs + h * D;
#pragma unroll
        for (int db = 0; db < D / 16; ++db) {
          half4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (half_t)(o[db][qb][r] * inv);
          *(half4*)(O + (long)row * p.o_rs + db * 16 + lg * 4) = v;
        }
      } else {
        // workspace [nsplit][B*H][Nq][D + 2] floats: O unnormalised, m (natural-log units), l -- the layout of mq_attn_fwd
        float* wr = ws + (((long)split * (p.B * p.H) + (long)b * p.H + h) * p.Nq + row) * (D + 2);
#pragma unroll
        for (int db = 0; db < D / 16; ++db)
#pragma unroll
          for (int r = 0; r < 4; ++r) wr[db * 16 + lg * 4 + r] = o[db][qb][r];
        if (lg == 0) { wr[D] = m_run[qb] * (1.f / RES_LOG2E); wr[D + 1] = l; }
      }
    }
  }
}

// merge the key-split partials: one thread per (b, h, row, 4 channels)
template <int D>
__global__ __launch_bounds__(256) void attn_chunked_combine_kernel(ResAttnParams p, const float* ws, int nsplit) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)p.B * p.H * p.Nq * (D / 4);
  if (t >= total) return;
  const int c4 = (int)(t % (D / 4));
  const long rw = t / (D / 4);                                      // (b * H + h) * Nq + row
  const int row = (int)(rw % p.Nq);
  const int bh = (int)(rw / p.Nq), b = bh / p.H, h = bh % p.H;
  const long stride = (long)p.B * p.H * p.Nq * (D + 2);
  const float* base = ws + rw * (D + 2);
  float mx = MQ_NEG_BIG;
  for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, base[s * stride + D]);
  float l = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < nsplit; ++s) {
    const float* w = base + s * stride;
    const float f = __expf(w[D] - mx);
    l += w[D + 1] * f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += w[c4 * 4 + j] * f;
  }
  const float inv = 1.f / l;
  half4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (half_t)(acc[j] * inv);
  *(half4*)(p.o + (long)b * p.o_bs + (long)row * p.o_rs + h * D + c4 * 4) = v;
}

template <int D, bool CLAMP>
static int launch_chunked_c(const ResAttnParams& p, float* ws, int nsplit, hipStream_t stream) {
  constexpr size_t smem = (size_t)2 * (RES_NKMAX * (D + 8) + D * (RES_NKMAX + 8)) * sizeof(half_t) + 4 * RES_NKMAX * sizeof(float);
  static MqOncePerDevice attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_chunked_kernel<D, CLAMP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set.done();
  }
  const int qtiles = (p.Nq + RES_BM - 1) / RES_BM;
  hipLaunchKernelGGL((attn_chunked_kernel<D, CLAMP>), dim3((unsigned)(qtiles * p.B * p.H * nsplit)), dim3(256), smem, stream, p, ws, nsplit);
  MQ_CHECK_LAUNCH();
  if (nsplit > 1) {
    const long total = (long)p.B * p.H * p.Nq * (D / 4);
    hipLaunchKernelGGL((attn_chunked_combine_kernel<D>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p, (const float*)ws, nsplit);
    MQ_CHECK_LAUNCH();
  }
  return 0;
}

// Same arguments as mq_attn_fwd (workspace: mq_attn_workspace_bytes(B, H, Nq, D, nsplit) when nsplit > 1); no qk_mask.
extern "C" int MQ_SYM(mq_attn_chunked_fwd)(const void* q, const void* k, const void* vt, void* o, const float* key_bias,
                                           const int* kv_len, void* workspace, int B, int H, int Nq, int Nk, int D,
                                           long q_bs, long q_rs, long q_hs, long k_bs, long k_rs, long k_hs,
                                           long vt_bs, long vt_rs, long vt_hs, long o_bs, long o_rs, long bias_bs, long bias_hs,
                                           float scale, float clamp, int nsplit, void* stream) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return 0;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > 1 && workspace == nullptr) return -2;
  if ((vt_rs % 8) || (q_rs % 8) || (k_rs % 8) || (q_hs % 8) || (k_hs % 8) || (vt_hs % 8) || (o_rs % 4)) return -3;
  ResAttnParams p;
  p.q = (const half_t*)q; p.k = (const half_t*)k; p.vt = (const half_t*)vt; p.o = (half_t*)o;
  p.key_bias = key_bias; p.kv_len = kv_len;
  p.qk_mask = nullptr; p.mask_bs = p.mask_hs = p.mask_rs = 0;
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
  p.q_bs = q_bs; p.q_rs = q_rs; p.q_hs = q_hs; p.k_bs = k_bs; p.k_rs = k_rs; p.k_hs = k_hs;
  p.vt_bs = vt_bs; p.vt_rs = vt_rs; p.vt_hs = vt_hs; p.bias_bs = bias_bs; p.bias_hs = bias_hs;
  p.o_bs = o_bs; p.o_rs = o_rs; p.scale = scale; p.clamp = clamp;
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const bool cl = clamp > 0.f;
  switch (D) {
    case 32: return cl ? launch_chunked_c<32, true>(p, ws, nsplit, s) : launch_chunked_c<32, false>(p, ws, nsplit, s);
    case 64: return cl ? launch_chunked_c<64, true>(p, ws, nsplit, s) : launch_chunked_c<64, false>(p, ws, nsplit, s);
    default: return -1;
  }
}

MQ_NAMESPACE_END
