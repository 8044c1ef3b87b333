// DyConv epilogue kernels for gfx950 (HBM-bound, NHWC fp16, fp32 statistics).
//
// Reference: modeling/rpn/vldyhead.py:148-152 (GroupNorm(16) after each DCNv2), :224 (bilinear up-sampling,
// align_corners=True, of the branch coming from level+1), :228-238 (scale attention: AvgPool -> 1x1 conv -> ReLU
// -> h_sigmoid per branch, mean over branches) and layers/dyrelu.py:78-112 (DYReLU: global pool -> FC 256->64->1024
// -> h_sigmoid -> max(a1 x + b1, a2 x + b2)).  In the reference (and in v1 of this repo) that is ~150 small
// launches per DyConv layer and ~10 full passes over every feature map; torch's fp16 NHWC bilinear kernel alone
// cost 14.5 ms / step.  Here:
//   mq_dyconv_stats   one pass over a branch's DCN output y[B,n,C]: per-(b,c) sum, sum of squares and the
//                     WEIGHTED sum that equals the spatial mean of the (optionally up-sampled) map;
//   mq_dyconv_coef    per (b, branch): GroupNorm mean/rstd from the channel sums, GN affine folded with the
//                     scale-attention scalar into  A[b,c], Bc[b,c]  (GN is affine per channel, so it commutes
//                     with the bilinear interpolation and with the spatial mean);
//   mq_dyconv_fuse    out[p,c] = sum_branches A*y^(p,c) + Bc  (y^ = y or its bilinear sample), plus the channel
//                     sums of `out` for DyReLU's pooling -- one read of every branch, one write;
//   mq_dyrelu_coef    the two tiny FCs + h_sigmoid -> per (b,c) a1,b1,a2,b2;
//   mq_dyrelu_apply   out = max(a1 x + b1, a2 x + b2) in place.
#include "common.h"

MQ_NAMESPACE_BEGIN

// ---------------------------------------------------------------------------------------------- stats
// part[b, blk, c, 0..2] = per-block (sum y, sum y^2, sum w_p y); wy/wx == nullptr -> w_p = 1/n.  Partials (not atomics)
// keep the reduction order fixed -> bitwise reproducible statistics.
__global__ __launch_bounds__(256) void dyconv_stats_kernel(const half_t* __restrict__ y, float* __restrict__ sums,
                                                           const float* __restrict__ wy, const float* __restrict__ wx,
                                                           int n, int W, int C, int rows_per_block) {
  __shared__ float red[8][256 * 3 / 8 * 8];     // [row group][c0 lanes(32) * 8 ch * 3]  (C == 256 path)
  const int b = blockIdx.y;
  const int lane_c = threadIdx.x % (C / 8), rg = threadIdx.x / (C / 8);
  const int nrg = 256 / (C / 8);
  const int c0 = lane_c * 8;
  const int p0 = blockIdx.x * rows_per_block;
  const int p1 = min(n, p0 + rows_per_block);
  float s1[8], s2[8], s3[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = s3[j] = 0.f;
  const float inv_n = 1.f / (float)n;
  for (int p = p0 + rg; p < p1; p += nrg) {
    half8 v = *(const half8*)(y + ((long)b * n + p) * C + c0);
    float w = wy ? wy[p / W] * wx[p % W] : inv_n;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = (float)v[j];
      s1[j] += f; s2[j] += f * f; s3[j] += w * f;
    }
  }
  float* r = &red[0][0] + (rg * (C / 8) + lane_c) * 24;
#pragma unroll
  for (int j = 0; j < 8; ++j) { r[j] = s1[j]; r[8 + j] = s2[j]; r[16 + j] = s3[j]; }
  __syncthreads();
  // thread t < C*3/... : reduce over row groups.  256 threads handle C*3 = 768 values -> 3 each
  for (int idx = threadIdx.x; idx < (C / 8) * 24; idx += 256) {
    int lc = idx / 24, k = idx % 24;
    float acc = 0.f;
    for (int g = 0; g < nrg; ++g) acc += (&red[0][0])[(g * (C / 8) + lc) * 24 + k];
    int c = lc * 8 + (k % 8), which = k / 8;
    sums[(((long)b * gridDim.x + blockIdx.x) * C + c) * 3 + which] = acc;
  }
}

extern "C" int MQ_SYM(mq_dyconv_stats)(const void* y, float* sums, const float* wy, const float* wx, int B, int n, int W, int C,
                               void* stream) {
  if (B <= 0 || n <= 0) return 0;
  if (C != 256) return -1;
  int rows = 256;
  dim3 grid((n + rows - 1) / rows, B);
  hipLaunchKernelGGL(dyconv_stats_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)y, sums, wy, wx, n, W, C, rows);
  MQ_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------- coef
// one block (C threads) per batch element.  coef[b, c, 0] = a*rstd*gamma, coef[b, c, 1] = a*(beta - mean*rstd*gamma)
// with a = h_sigmoid(relu(attn_w . pooled + attn_b)) / nbranches, pooled_c = GN affine of the weighted mean.
__global__ void dyconv_coef_kernel(const float* __restrict__ part, int nblk, const half_t* __restrict__ gamma,
                                   const half_t* __restrict__ beta, const float* __restrict__ attn_w,
                                   const float* __restrict__ attn_b, float* __restrict__ coef, int n, int C, int G,
                                   float eps, float inv_nbr) {
  __shared__ float cs[256], css[256];
  __shared__ float gs[64], gss[64];
  __shared__ float dotp[256];
  const int b = blockIdx.x, c = threadIdx.x;
  float s[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < nblk; ++k) {                      // fixed order
    const float* pp = part + (((long)b * nblk + k) * C + c) * 3;
    s[0] += pp[0]; s[1] += pp[1]; s[2] += pp[2];
  }
  const int cpg = C / G;
  cs[c] = s[0]; css[c] = s[1];
  __syncthreads();
  if (c < G) {
    float a0 = 0.f, a1 = 0.f;
    for (int j = 0; j < cpg; ++j) { a0 += cs[c * cpg + j]; a1 += css[c * cpg + j]; }
    gs[c] = a0; gss[c] = a1;
  }
  __syncthreads();
  const float cnt = (float)n * cpg;
  const float mean = gs[c / cpg] / cnt;
  const float var = fmaxf(gss[c / cpg] / cnt - mean * mean, 0.f);
  const float rstd = rsqrtf(var + eps);
  const float sc = rstd * (float)gamma[c];
  const float sh = (float)beta[c] - mean * sc;
  dotp[c] = attn_w[c] * (sc * s[2] + sh);          // s[2] = weighted spatial mean of y
  __syncthreads();
  for (int o = C / 2; o > 0; o >>= 1) {
    if (c < o) dotp[c] += dotp[c + o];
    __syncthreads();
  }
  float a = fmaxf(dotp[0] + attn_b[0], 0.f);
  a = fminf(fmaxf(a + 3.f, 0.f), 6.f) / 6.f * inv_nbr;
  coef[((long)b * C + c) * 2 + 0] = a * sc;
  coef[((long)b * C + c) * 2 + 1] = a * sh;
}

extern "C" int MQ_SYM(mq_dyconv_coef)(const float* sums, const void* gamma, const void* beta, const float* attn_w,
                              const float* attn_b, float* coef, int B, int n, int nblk, int C, int G, float eps,
                              int nbranches, void* stream) {
  if (B <= 0) return 0;
  if (C != 256 || G > 64 || C % G) return -1;
  hipLaunchKernelGGL(dyconv_coef_kernel, dim3(B), dim3(C), 0, (hipStream_t)stream, sums, nblk > 0 ? nblk : (n + 255) / 256, (const half_t*)gamma,
                     (const half_t*)beta, attn_w, attn_b, coef, n, C, G, eps, 1.f / (float)nbranches);
  MQ_CHECK_LAUNCH();
  return 0;
}

// Grouped form: the coefficient sets of ALL branches of a DyConv layer (13) in one launch -- grid (B, branches), 1024
// threads = 4 partial reducers x 256 channels with independent loads in flight (the single-branch kernel walks the
// per-tile partials of a P3 branch, 143 of them, as one dependent chain: 13 us x 78 launches per forward).
struct CoefBranch { const float* part; const half_t* gamma; const half_t* beta; float* coef; int nblk, n; float inv_nbr; int pad; };
struct CoefGroup { CoefBranch br[16]; const float* attn_w; const float* attn_b; int C, G; float eps; int nbr; };

__global__ __launch_bounds__(1024) void dyconv_coef_group_kernel(CoefGroup g) {
  __shared__ float red[4][256][3];
  __shared__ float cs[256], css[256];
  __shared__ float gs[64], gss[64];
  __shared__ float dotp[256];
  const CoefBranch br = g.br[blockIdx.y];
  const int b = blockIdx.x, c = threadIdx.x & 255, q = threadIdx.x >> 8, C = g.C;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const float* base = br.part + ((long)b * br.nblk * C + c) * 3;
  // fixed order per reducer (k = q, q + 4, ...), eight partials' loads in flight at a time: one load per iteration was a chain of 36
  // L2 round trips on the P3 branches (143 partials) -- 38 us per launch between the DCNv2 launch and the epilogue of every layer
  int k = q;
  for (; k + 28 < br.nblk; k += 32) {
    float v[8][3];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* pp = base + (long)(k + 4 * u) * C * 3;
      v[u][0] = pp[0]; v[u][1] = pp[1]; v[u][2] = pp[2];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s0 += v[u][0]; s1 += v[u][1]; s2 += v[u][2]; }
  }
  for (; k < br.nblk; k += 4) {
    const float* pp = base + (long)k * C * 3;
    s0 += pp[0]; s1 += pp[1]; s2 += pp[2];
  }
  red[q][c][0] = s0; red[q][c][1] = s1; red[q][c][2] = s2;
  __syncthreads();
  if (q != 0) {
    // reducers 1..3 only feed reducer 0; they still take part in every barrier below
  }
  float s[3] = {0.f, 0.f, 0.f};
  if (q == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { s[0] += red[j][c][0]; s[1] += red[j][c][1]; s[2] += red[j][c][2]; }
    cs[c] = s[0]; css[c] = s[1];
  }
  __syncthreads();
  const int cpg = C / g.G;
  if (threadIdx.x < g.G) {
    float a0 = 0.f, a1 = 0.f;
    for (int j = 0; j < cpg; ++j) { a0 += cs[threadIdx.x * cpg + j]; a1 += css[threadIdx.x * cpg + j]; }
    gs[threadIdx.x] = a0; gss[threadIdx.x] = a1;
  }
  __syncthreads();
  const float cnt = (float)br.n * cpg;
  const float mean = gs[c / cpg] / cnt;
  const float var = fmaxf(gss[c / cpg] / cnt - mean * mean, 0.f);
  const float rstd = rsqrtf(var + g.eps);
  const float sc = rstd * (float)br.gamma[c];
  const float sh = (float)br.beta[c] - mean * sc;
  if (q == 0) dotp[c] = g.attn_w[c] * (sc * s[2] + sh);          // s[2] = weighted spatial mean of y
  __syncthreads();
  for (int o = C / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) dotp[threadIdx.x] += dotp[threadIdx.x + o];
    __syncthreads();
  }
  if (q == 0) {
    float a = fmaxf(dotp[0] + g.attn_b[0], 0.f);
    a = fminf(fmaxf(a + 3.f, 0.f), 6.f) / 6.f * br.inv_nbr;
    br.coef[((long)b * C + c) * 2 + 0] = a * sc;
    br.coef[((long)b * C + c) * 2 + 1] = a * sh;
  }
}

struct mq_coef_branch {          // mirrors include/mqdet_hip.h
  const float* sums; const void* gamma; const void* beta; float* coef; int nblk, n, nbranches, reserved;
};

extern "C" int MQ_SYM(mq_dyconv_coef_group)(const mq_coef_branch* br, int nbr, const float* attn_w, const float* attn_b, int B, int C,
                                    int G, float eps, void* stream) {
  if (B <= 0 || nbr <= 0) return 0;
  if (C != 256 || G > 64 || C % G || nbr > 16) return -1;
  CoefGroup g;
  for (int i = 0; i < nbr; ++i) {
    g.br[i].part = br[i].sums; g.br[i].gamma = (const half_t*)br[i].gamma; g.br[i].beta = (const half_t*)br[i].beta;
    g.br[i].coef = br[i].coef; g.br[i].nblk = br[i].nblk; g.br[i].n = br[i].n; g.br[i].inv_nbr = 1.f / (float)br[i].nbranches;
    g.br[i].pad = 0;
  }
  g.attn_w = attn_w; g.attn_b = attn_b; g.C = C; g.G = G; g.eps = eps; g.nbr = nbr;
  hipLaunchKernelGGL(dyconv_coef_group_kernel, dim3(B, nbr), dim3(1024), 0, (hipStream_t)stream, g);
  MQ_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------- fuse
struct FuseBranch {
  const half_t* y;      // [B, hs*ws, C]
  const float* coef;    // [B, C, 2]
  int hs, ws;           // source dims; (hs, ws) == (H, W) -> direct, else bilinear align_corners=True
};
struct FuseParams {
  FuseBranch br[3];
  int nbr;
  half_t* out;          // [B, H*W, C], batch stride out_bs elements (a level's slice of the [B, N, C] token buffer)
  long out_bs;
  float* pool;          // [B, nblk, C] per-block partial sums of out (fixed-order reduction in mq_dyrelu_coef)
  int B, H, W, C, rows_per_block;
};

// ND = branches at the output resolution (read directly), HB = one more branch at a coarser resolution (bilinear,
// align_corners = True) -- host order: direct branches first.  Every thread walks its positions U = 4 at a time and issues
// ALL loads of the four positions (<= 24 x 16 B) before the first use: the first version consumed each position's loads
// before issuing the next one's -- one dependent chain per thread, 1.3 TB/s on an HBM-bound pass.
// (body shared by the per-level kernel and the grouped one: b = image, blk / nblk = this workgroup's 128-position block of the level)
// U = positions whose loads are in flight together (the visiting order of a thread's positions, hence every sum, does not depend on it)
template <int ND, bool HB, int U = 4>
__device__ __forceinline__ void dyconv_fuse_body(const FuseParams& p, int b, int blk, int nblk, float* red) {
  constexpr int NBR = ND + (HB ? 1 : 0);
  const int cpt = p.C / 8;
  const int lane_c = threadIdx.x % cpt, rg = threadIdx.x / cpt, nrg = 256 / cpt;
  const int c0 = lane_c * 8;
  const int n = p.H * p.W;
  const int p0 = blk * p.rows_per_block, p1 = min(n, p0 + p.rows_per_block);
  float A[NBR][8], Bc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) Bc[j] = 0.f;
#pragma unroll
  for (int k = 0; k < NBR; ++k) {
    const float* cf = p.br[k].coef + ((long)b * p.C + c0) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) { A[k][j] = cf[j * 2]; Bc[j] += cf[j * 2 + 1]; }
  }
  const half_t* yd[ND > 0 ? ND : 1];
#pragma unroll
  for (int k = 0; k < ND; ++k) yd[k] = p.br[k].y + (long)b * n * p.C + c0;
  const FuseBranch& bb = p.br[NBR - 1];
  const half_t* yb = HB ? bb.y + (long)b * bb.hs * bb.ws * p.C + c0 : nullptr;
  const float ry = (HB && p.H > 1) ? (float)(bb.hs - 1) / (float)(p.H - 1) : 0.f;
  const float rx = (HB && p.W > 1) ? (float)(bb.ws - 1) / (float)(p.W - 1) : 0.f;
  float ps[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ps[j] = 0.f;
  for (int pos0 = p0 + rg; pos0 < p1; pos0 += nrg * U) {
    half8 vd[U][ND > 0 ? ND : 1], vb[U][4];
    float ly[U], lx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pos = min(pos0 + u * nrg, p1 - 1);
#pragma unroll
      for (int k = 0; k < ND; ++k) vd[u][k] = *(const half8*)(yd[k] + (long)pos * p.C);
      if constexpr (HB) {
        const int oy = pos / p.W, ox = pos - oy * p.W;
        // same expression as the reference's F.interpolate(align_corners=True) source index: oy * (hs - 1) / (H - 1)
        const float sy = p.H > 1 ? (float)oy * (float)(bb.hs - 1) / (float)(p.H - 1) : 0.f;
        const float sx = p.W > 1 ? (float)ox * (float)(bb.ws - 1) / (float)(p.W - 1) : 0.f;
        const int y0 = min((int)sy, bb.hs - 1), x0 = min((int)sx, bb.ws - 1);
        const int y1 = min(y0 + 1, bb.hs - 1), x1 = min(x0 + 1, bb.ws - 1);
        ly[u] = sy - (float)y0; lx[u] = sx - (float)x0;
        vb[u][0] = *(const half8*)(yb + ((long)y0 * bb.ws + x0) * p.C);
        vb[u][1] = *(const half8*)(yb + ((long)y0 * bb.ws + x1) * p.C);
        vb[u][2] = *(const half8*)(yb + ((long)y1 * bb.ws + x0) * p.C);
        vb[u][3] = *(const half8*)(yb + ((long)y1 * bb.ws + x1) * p.C);
      }
    }
    (void)ry; (void)rx;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pos = pos0 + u * nrg;
      if (pos >= p1) break;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = Bc[j];
#pragma unroll
      for (int k = 0; k < ND; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += A[k][j] * (float)vd[u][k][j];
      if constexpr (HB) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = (1.f - ly[u]) * ((1.f - lx[u]) * (float)vb[u][0][j] + lx[u] * (float)vb[u][1][j]) +
                          ly[u] * ((1.f - lx[u]) * (float)vb[u][2][j] + lx[u] * (float)vb[u][3][j]);
          acc[j] += A[NBR - 1][j] * v;
        }
      }
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) { o[j] = (half_t)acc[j]; ps[j] += (float)o[j]; }
      *(half8*)(p.out + (long)b * p.out_bs + (long)pos * p.C + c0) = o;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[(rg * cpt + lane_c) * 8 + j] = ps[j];
  __syncthreads();
  for (int c = threadIdx.x; c < p.C; c += 256) {
    float acc = 0.f;
    for (int g = 0; g < nrg; ++g) acc += red[(g * cpt + c / 8) * 8 + (c % 8)];
    p.pool[((long)b * nblk + blk) * p.C + c] = acc;
  }
}

template <int ND, bool HB>
__global__ __launch_bounds__(256) void dyconv_fuse_kernel(FuseParams p) {
  __shared__ float red[256 * 8];
  dyconv_fuse_body<ND, HB>(p, blockIdx.y, blockIdx.x, gridDim.x, red);
}

// Grouped form: the epilogues of ALL levels of a DyConv layer in one launch (KERNELS["DYCONV_EPILOGUE_GROUPED"]).  Per level they were a
// launch each on five streams (P5 .. P7: 72 / 24 / 8 workgroups) behind a fork / join; the blocks of all levels form one work list,
// the branch mix of a level (1 .. 3 direct branches, one bilinear) picks the body.
static constexpr int FUSE_MAX_LEVELS = 8;
struct FuseGroup {
  FuseParams lv[FUSE_MAX_LEVELS];
  int first_block[FUSE_MAX_LEVELS + 1];
  int nblk[FUSE_MAX_LEVELS], kind[FUSE_MAX_LEVELS];          // kind = 2 * nd + (bilinear branch ? 1 : 0)
  int n;
};
// (split-precise build: the fp32 loads in flight are twice the registers -- 173 spilled VGPRs at three waves per SIMD; two per SIMD there)
#if defined(MQ_F32)
#define MQ_FUSE_GROUP_WAVES 2
#else
#define MQ_FUSE_GROUP_WAVES 3
#endif
__global__ __launch_bounds__(256, MQ_FUSE_GROUP_WAVES) void dyconv_fuse_group_kernel(FuseGroup g) {
  __shared__ float red[256 * 8];
  int L = 0;
  while (L + 1 < g.n && (int)blockIdx.x >= g.first_block[L + 1]) ++L;
  const FuseParams& p = g.lv[L];
  const int t = blockIdx.x - g.first_block[L], nblk = g.nblk[L];
  const int b = t / nblk, blk = t - b * nblk;
  switch (g.kind[L]) {
    case 2: dyconv_fuse_body<1, false>(p, b, blk, nblk, red); break;
    case 4: dyconv_fuse_body<2, false>(p, b, blk, nblk, red); break;
    case 6: dyconv_fuse_body<3, false>(p, b, blk, nblk, red); break;
    case 1: dyconv_fuse_body<0, true>(p, b, blk, nblk, red); break;
    case 3: dyconv_fuse_body<1, true>(p, b, blk, nblk, red); break;
    default: dyconv_fuse_body<2, true, 3>(p, b, blk, nblk, red); break;     // U = 3: <= 168 VGPRs, three waves per SIMD like the P3 body
  }
}

// direct branches first, the (single) coarser one last; false: more than one coarser branch
static bool fuse_params(FuseParams& p, const FuseBranch* in, int nbranches, void* out, long out_bs, float* pool, int B, int H, int W, int C,
                        int& nd, int& nbil) {
  nd = nbil = 0;
  for (int k = 0; k < nbranches; ++k)
    if (in[k].hs == H && in[k].ws == W) p.br[nd++] = in[k];
  for (int k = 0; k < nbranches; ++k)
    if (!(in[k].hs == H && in[k].ws == W)) { p.br[nd + nbil] = in[k]; ++nbil; }
  for (int k = nbranches; k < 3; ++k) p.br[k] = in[0];
  p.nbr = nbranches; p.out = (half_t*)out; p.out_bs = out_bs; p.pool = pool; p.B = B; p.H = H; p.W = W; p.C = C;
  p.rows_per_block = 128;
  return nbil <= 1;
}

extern "C" int MQ_SYM(mq_dyconv_fuse)(const void* y0, const float* coef0, int hs0, int ws0, const void* y1, const float* coef1,
                              int hs1, int ws1, const void* y2, const float* coef2, int hs2, int ws2, int nbranches,
                              void* out, long out_bs, float* pool, int B, int H, int W, int C, void* stream) {
  if (B <= 0) return 0;
  if (C != 256 || nbranches < 1 || nbranches > 3) return -1;
  FuseParams p;
  const FuseBranch in[3] = {{(const half_t*)y0, coef0, hs0, ws0}, {(const half_t*)y1, coef1, hs1, ws1}, {(const half_t*)y2, coef2, hs2, ws2}};
  int nd = 0, nbil = 0;
  if (!fuse_params(p, in, nbranches, out, out_bs, pool, B, H, W, C, nd, nbil)) return -2;
  dim3 grid((H * W + p.rows_per_block - 1) / p.rows_per_block, B);
  hipStream_t st = (hipStream_t)stream;
  if (nbil == 0) {
    if (nd == 1) hipLaunchKernelGGL((dyconv_fuse_kernel<1, false>), grid, dim3(256), 0, st, p);
    else if (nd == 2) hipLaunchKernelGGL((dyconv_fuse_kernel<2, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((dyconv_fuse_kernel<3, false>), grid, dim3(256), 0, st, p);
  } else {
    if (nd == 0) hipLaunchKernelGGL((dyconv_fuse_kernel<0, true>), grid, dim3(256), 0, st, p);
    else if (nd == 1) hipLaunchKernelGGL((dyconv_fuse_kernel<1, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((dyconv_fuse_kernel<2, true>), grid, dim3(256), 0, st, p);
  }
  MQ_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------- DyReLU
// one block (256 threads) per batch element: y = pool / n -> fc0 (C -> C/4) ReLU -> fc2 (C/4 -> 4C) -> h_sigmoid
// coef[b, 0..3, c] = a1, b1, a2, b2  (lambda_a = 2, init_a = (1, 0), init_b = (0, 0))
__device__ __forceinline__ void dyrelu_coef_body(const float* __restrict__ pool, int nblk, const half_t* __restrict__ w0,
                                                 const half_t* __restrict__ b0, const half_t* __restrict__ w2,
                                                 const half_t* __restrict__ b2, float* __restrict__ coef, int n, int C, int b,
                                                 float* yv, float* hv, float (*part)[256]) {
  const int t = threadIdx.x & 255, grp = threadIdx.x >> 8;
  const int S = C / 4;                                   // C == 256, S == 64 (checked by the host wrapper)
  // spatial mean: fixed-order sum of the per-block partials.  The P3 level has 132 of them per image: one dependent chain of 17
  // rounds of 8 loads was 42 us (the launch sits between the fuse kernel and the next LayerNorm, on the critical path); four
  // row groups each take a contiguous quarter (8 loads in flight each) and are combined in group order.
  {
    const int q = (nblk + 3) / 4, k0 = grp * q, k1 = min(nblk, k0 + q);
    float acc0 = 0.f;
    const float* pb = pool + (long)b * nblk * C + t;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pb[(long)(k + u) * C];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc0 += v[u];
    }
    for (; k < k1; ++k) acc0 += pb[(long)k * C];
    part[grp][t] = acc0;
  }
  __syncthreads();
  if (grp == 0) yv[t] = (((part[0][t] + part[1][t]) + part[2][t]) + part[3][t]) / (float)n;
  __syncthreads();
  if (grp == 0) {  // fc.0 (C -> S) + ReLU: 4 lanes per output, 16-byte weight loads, fixed-order combine
    const int o = t >> 2, q = t & 3;
    const half_t* wr = w0 + (long)o * C + q * (C / 4);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {                          // C / 4 = 64 weights = 8 x half8
      const half8 w = *(const half8*)(wr + i * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) a += (float)w[j] * yv[q * (C / 4) + i * 8 + j];
    }
    a += __shfl_xor(a, 1);
    a += __shfl_xor(a, 2);
    if (q == 0) hv[o] = fmaxf(a + (float)b0[o], 0.f);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 4 * C; o += 1024) {         // fc.2 (S -> 4C) + h_sigmoid, 16-byte weight loads: one output per thread
    float acc = (float)b2[o];
    const half_t* wr = w2 + (long)o * S;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const half8 w = *(const half8*)(wr + i * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += (float)w[j] * hv[i * 8 + j];
    }
    float hs = fminf(fmaxf(acc + 3.f, 0.f), 6.f) / 6.f;
    int which = o / C, c = o % C;
    float v = which == 0 ? (hs - 0.5f) * 2.f + 1.f : (which == 2 ? (hs - 0.5f) * 2.f : hs - 0.5f);
    coef[((long)b * 4 + which) * C + c] = v;
  }
}

__global__ __launch_bounds__(1024) void dyrelu_coef_kernel(const float* __restrict__ pool, int nblk,
                                                           const half_t* __restrict__ w0,
                                                           const half_t* __restrict__ b0, const half_t* __restrict__ w2,
                                                           const half_t* __restrict__ b2, float* __restrict__ coef, int n,
                                                           int C) {
  __shared__ float yv[256], hv[64], part[4][256];
  dyrelu_coef_body(pool, nblk, w0, b0, w2, b2, coef, n, C, blockIdx.x, yv, hv, part);
}

// all levels of a layer: grid (B, levels) -- the five launches of 8 workgroups each were a chain link of ~20 us apiece on five streams
struct ReluGroup {
  const float* pool[FUSE_MAX_LEVELS]; float* coef[FUSE_MAX_LEVELS];
  int nblk[FUSE_MAX_LEVELS], n[FUSE_MAX_LEVELS];
  const half_t* w0; const half_t* b0; const half_t* w2; const half_t* b2;
  int C;
};
__global__ __launch_bounds__(1024) void dyrelu_coef_group_kernel(ReluGroup g) {
  __shared__ float yv[256], hv[64], part[4][256];
  const int L = blockIdx.y;
  dyrelu_coef_body(g.pool[L], g.nblk[L], g.w0, g.b0, g.w2, g.b2, g.coef[L], g.n[L], g.C, blockIdx.x, yv, hv, part);
}

struct mq_fuse_level {          // mirrors include/mqdet_hip.h
  const void* y[3]; const float* coef[3]; int hs[3], ws[3];
  int nbranches, H, W, reserved;
  void* out; long out_bs; float* pool; float* relu_coef;
};

// The epilogue of a DyConv layer for all pyramid levels in two launches: mq_dyconv_fuse of every level (one work list), then
// mq_dyrelu_coef of every level.  levels[i]: the arguments of those two entry points (pool [B, ceil(H*W/128), C] workspace,
// relu_coef [B,4,C] out).  C == 256, <= 8 levels.
extern "C" int MQ_SYM(mq_dyconv_epilogue_group)(const mq_fuse_level* levels, int nl, const void* w0, const void* b0, const void* w2,
                                                const void* b2, int B, int C, void* stream) {
  if (B <= 0 || nl <= 0) return 0;
  if (C != 256 || nl > FUSE_MAX_LEVELS) return -1;
  FuseGroup g;
  ReluGroup r;
  g.n = 0;
  g.first_block[0] = 0;
  for (int i = 0; i < nl; ++i) {
    const mq_fuse_level& a = levels[i];
    if (a.nbranches < 1 || a.nbranches > 3 || a.H <= 0 || a.W <= 0) return -1;
    FuseBranch in[3];
    for (int k = 0; k < 3; ++k) in[k] = FuseBranch{(const half_t*)a.y[k], a.coef[k], a.hs[k], a.ws[k]};
    int nd = 0, nbil = 0;
    if (!fuse_params(g.lv[g.n], in, a.nbranches, a.out, a.out_bs, a.pool, B, a.H, a.W, C, nd, nbil)) return -2;
    g.kind[g.n] = 2 * nd + nbil;
    g.nblk[g.n] = (a.H * a.W + 127) / 128;
    g.first_block[g.n + 1] = g.first_block[g.n] + B * g.nblk[g.n];
    r.pool[g.n] = a.pool; r.coef[g.n] = a.relu_coef; r.nblk[g.n] = g.nblk[g.n]; r.n[g.n] = a.H * a.W;
    ++g.n;
  }
  for (int i = g.n; i < FUSE_MAX_LEVELS; ++i) {
    g.lv[i] = g.lv[0]; g.nblk[i] = g.nblk[0]; g.kind[i] = g.kind[0]; g.first_block[i + 1] = g.first_block[g.n];
    r.pool[i] = r.pool[0]; r.coef[i] = r.coef[0]; r.nblk[i] = r.nblk[0]; r.n[i] = r.n[0];
  }
  r.w0 = (const half_t*)w0; r.b0 = (const half_t*)b0; r.w2 = (const half_t*)w2; r.b2 = (const half_t*)b2; r.C = C;
  hipLaunchKernelGGL(dyconv_fuse_group_kernel, dim3((unsigned)g.first_block[g.n]), dim3(256), 0, (hipStream_t)stream, g);
  MQ_CHECK_LAUNCH();
  hipLaunchKernelGGL(dyrelu_coef_group_kernel, dim3(B, g.n), dim3(1024), 0, (hipStream_t)stream, r);
  MQ_CHECK_LAUNCH();
  return 0;
}

extern "C" int MQ_SYM(mq_dyrelu_coef)(const float* pool, const void* w0, const void* b0, const void* w2, const void* b2,
                              float* coef, int B, int n, int C, void* stream) {
  if (B <= 0) return 0;
  if (C != 256) return -1;
  hipLaunchKernelGGL(dyrelu_coef_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, pool, (n + 127) / 128, (const half_t*)w0,
                     (const half_t*)b0, (const half_t*)w2, (const half_t*)b2, coef, n, C);
  MQ_CHECK_LAUNCH();
  return 0;
}

__global__ __launch_bounds__(256) void dyrelu_apply_kernel(half_t* __restrict__ x, const float* __restrict__ coef, long n,
                                                           int C, long x_bs) {
  const int b = blockIdx.y;
  const int cpt = C / 8;
  const long total = n * cpt;
  const float* cf = coef + (long)b * 4 * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c0 = (int)(i % cpt) * 8;
    half_t* px = x + (long)b * x_bs + (i / cpt) * C + c0;
    half8 v = *(const half8*)px;
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = (float)v[j];
      o[j] = (half_t)fmaxf(f * cf[c0 + j] + cf[C + c0 + j], f * cf[2 * C + c0 + j] + cf[3 * C + c0 + j]);
    }
    *(half8*)px = o;
  }
}

extern "C" int MQ_SYM(mq_dyrelu_apply)(void* x, const float* coef, int B, int n, int C, long x_bs, void* stream) {
  if (B <= 0 || n <= 0) return 0;
  if (C % 8) return -1;
  long total = (long)n * (C / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dyrelu_apply_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, (half_t*)x, coef,
                     (long)n, C, x_bs);
  MQ_CHECK_LAUNCH();
  return 0;
}

// ---- FPN top-down step (fpn.py:82-95): dst[b, h, w, :] += src[b, nearest(h), nearest(w), :] in place, NHWC 16-bit.
// F.interpolate(mode="nearest", size=...) + add were three passes (the up-sampled tensor written, read back with the lateral, the sum
// written); here the lateral is read and written once and the coarse level is read through the cache.  Source index as ATen's
// nearest_neighbor_compute_source_index: min(floor(dst * (float)(in / out)), in - 1); the sum is rounded once, as torch's add does.
__global__ __launch_bounds__(256) void add_upsample_nearest_kernel(half_t* __restrict__ dst, const half_t* __restrict__ src, int H, int W,
                                                                   int Hc, int Wc, int C, float sh, float sw) {
  const int b = blockIdx.y, cpt = C / 8;
  const long total = (long)H * W * cpt;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % cpt);
    const long pix = i / cpt;
    const int w = (int)(pix % W), h = (int)(pix / W);
    const int hs = min((int)floorf((float)h * sh), Hc - 1), wsrc = min((int)floorf((float)w * sw), Wc - 1);
    half_t* pd = dst + (((long)b * H + h) * W + w) * C + ch * 8;
    const half8 a = *(const half8*)pd;
    const half8 u = *(const half8*)(src + (((long)b * Hc + hs) * Wc + wsrc) * C + ch * 8);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)a[j] + (float)u[j]);
    *(half8*)pd = o;
  }
}

extern "C" int MQ_SYM(mq_add_upsample_nearest)(void* dst, const void* src, int B, int H, int W, int Hc, int Wc, int C, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  if (C % 8 || Hc < 1 || Wc < 1) return -1;
  long blocks = ((long)H * W * (C / 8) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_upsample_nearest_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, (half_t*)dst,
                     (const half_t*)src, H, W, Hc, Wc, C, (float)Hc / (float)H, (float)Wc / (float)W);
  MQ_CHECK_LAUNCH();
  return 0;
}

// ---- pooled FPN tokens of the GCP pre-select (generalized_vl_rcnn_new.py:291-293): torch.cat([F.avg_pool2d(f, 2) tokens of every level], 1).
// out[b, first[l] + y * (W_l / 2) + x, :] = mean of the 2 x 2 window of level l (floor sizes: a last odd row / column is dropped), summed in
// fp32 in the window's row-major order and rounded once, like ATen's NHWC average pool.  One launch instead of five pools + a concat.
struct PoolLevels {
  const half_t* x[8];
  long x_bs[8];
  int H[8], W[8], first[9];          // first[l]: first output token of level l; first[n]: tokens per image
  int n;
};
__global__ __launch_bounds__(256) void pool2x2_tokens_kernel(PoolLevels g, half_t* __restrict__ out, int C) {
  const int b = blockIdx.y, cpt = C / 8;
  const long total = (long)g.first[g.n] * cpt;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % cpt), tok = (int)(i / cpt);
    int l = 0;
    while (l + 1 < g.n && tok >= g.first[l + 1]) ++l;
    const int t = tok - g.first[l], w2 = g.W[l] >> 1;
    const int y = t / w2, x = t - y * w2;
    const half_t* p = g.x[l] + (long)b * g.x_bs[l] + ((long)(2 * y) * g.W[l] + 2 * x) * C + ch * 8;
    const half8 a = *(const half8*)p, c = *(const half8*)(p + C);
    const half8 d = *(const half8*)(p + (long)g.W[l] * C), e = *(const half8*)(p + (long)g.W[l] * C + C);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)((((float)a[j] + (float)c[j]) + (float)d[j] + (float)e[j]) * 0.25f);
    *(half8*)(out + ((long)b * g.first[g.n] + tok) * C + ch * 8) = o;
  }
}

struct mq_conv_level {          // mirrors include/mqdet_hip.h (as in conv_small3.hip)
  const void* x; float* out; long x_bs; int H, W;
};

extern "C" int MQ_SYM(mq_pool2x2_tokens_fwd)(const mq_conv_level* levels, int nl, void* out, int B, int C, void* stream) {
  if (B <= 0 || nl <= 0) return 0;
  if (nl > 8 || C % 8) return -1;
  PoolLevels g;
  g.n = 0;
  g.first[0] = 0;
  for (int i = 0; i < nl; ++i) {
    const int h2 = levels[i].H / 2, w2 = levels[i].W / 2;
    if (h2 <= 0 || w2 <= 0) continue;                       // a level smaller than the window has no tokens (avg_pool2d would refuse it)
    g.x[g.n] = (const half_t*)levels[i].x; g.x_bs[g.n] = levels[i].x_bs; g.H[g.n] = levels[i].H; g.W[g.n] = levels[i].W;
    g.first[g.n + 1] = g.first[g.n] + h2 * w2;
    ++g.n;
  }
  if (g.n == 0) return 0;
  long blocks = ((long)g.first[g.n] * (C / 8) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pool2x2_tokens_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, g, (half_t*)out, C);
  MQ_CHECK_LAUNCH();
  return 0;
}

MQ_NAMESPACE_END
