// mq_layernorm_fwd: row LayerNorm for gfx950, fp16 in/out, fp32 statistics (two-pass in registers), optional
// transposed second output.
//
// Every LayerNorm of the MQ-GLIP forward goes through here: Swin norm1/norm2/patch-merging/out norms
// (backbone/swint.py:198,240,281,611), BERT / GCP / VLFuse norms.  torch's LayerNorm kernel runs the C = 96 / 192
// rows of Swin stage 1-2 at ~0.4 TB/s (profiles/r01_call8: 534 us for a 103 MB tensor); here a row is owned by
// LPR = 16 / 32 / 64 lanes (C/8 sixteen-byte chunks, several rows per wave for small C) so every access is a
// coalesced 16-byte vector, ~HBM speed.  With `yt` != NULL (VLFuse: the text->image attention needs LN(v)^T as its
// V^T operand) the normalised tile is also transposed through LDS and written as yt[b, c, n] -- this replaces a
// separate 92 MB transpose copy per fusion layer.  With `res` != NULL the row is x + res (rounded to fp16 first, exactly
// what a separate elementwise add would hand to LayerNorm), optionally written out as `xsum`: the residual adds of the
// Swin blocks (swint.py:236,240) and of the BERT output blocks (LayerNorm(dense(h) + input)) cost no extra pass.
#include "common.h"

template <int LPR>
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, const half_t* __restrict__ res,
                                                        const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                                                        half_t* __restrict__ y, half_t* __restrict__ xsum,
                                                        half_t* __restrict__ yt, long rows, int C, float eps,
                                                        long rows_per_batch, long yt_ld, int RPB) {
  constexpr int ROWS_PER_PASS = 256 / LPR;          // RPB = rows per block: 64 (big inputs / transposed output) or one pass
  extern __shared__ __attribute__((aligned(16))) half_t tile[];      // [RPB][C + 8] when yt
  const int sub = threadIdx.x % LPR, rg = threadIdx.x / LPR;
  const int nch = C / 8;
  const long r0 = (long)blockIdx.x * RPB;
  constexpr int MAXC = 4;                          // chunks per lane (C <= 8 * LPR * MAXC)
  for (int rr = rg; rr < RPB; rr += ROWS_PER_PASS) {
    const long row = r0 + rr;
    const bool ok = row < rows;
    half8 v[MAXC];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int ch = sub + k * LPR;
      v[k] = zero8();
      if (ok && ch < nch) {
        v[k] = *(const half8*)(x + row * C + ch * 8);
        if (res) {
          const half8 r = *(const half8*)(res + row * C + ch * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[k][j] = (half_t)((float)v[k][j] + (float)r[j]);
          if (xsum) *(half8*)(xsum + row * C + ch * 8) = v[k];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)v[k][j];
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int ch = sub + k * LPR;
      if (ch < nch) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { float d = (float)v[k][j] - mean; q += d * d; }
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int ch = sub + k * LPR;
      if (ch < nch) {
        half8 g = *(const half8*)(gamma + ch * 8), bb = *(const half8*)(beta + ch * 8), o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)(((float)v[k][j] - mean) * rstd * (float)g[j] + (float)bb[j]);
        if (ok) *(half8*)(y + row * C + ch * 8) = o;
        if (yt) *(half8*)(tile + rr * (C + 8) + ch * 8) = o;
      }
    }
  }
  if (yt) {
    __syncthreads();
    // yt[b, c, n0 .. n0+63]: one 16-byte store per (channel, 8 consecutive rows); a block never straddles a batch
    // element when rows_per_batch % 64 == 0, otherwise rows are handled one by one
    for (int t = threadIdx.x; t < C * (64 / 8); t += 256) {
      const int c = t / (64 / 8), g8 = t % (64 / 8);
      const long row = r0 + g8 * 8;
      if (row >= rows) continue;
      const long b = row / rows_per_batch, n = row % rows_per_batch;
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = tile[(g8 * 8 + j) * (C + 8) + c];
      half_t* dst = yt + (b * C + c) * yt_ld + n;
      if (n + 8 <= rows_per_batch && row + 8 <= rows && (((size_t)dst) & 15) == 0) {
        *(half8*)dst = o;
      } else {
        for (int j = 0; j < 8 && row + j < rows; ++j) {
          const long bj = (row + j) / rows_per_batch, nj = (row + j) % rows_per_batch;
          yt[(bj * C + c) * yt_ld + nj] = o[j];
        }
      }
    }
  }
}

extern "C" int mq_layernorm_fwd(const void* x, const void* res, const void* gamma, const void* beta, void* y, void* xsum,
                                void* yt, long rows, int C, float eps, long rows_per_batch, long yt_ld, void* stream) {
  if (rows <= 0) return 0;
  if (C % 8 || C > 2048) return -1;
  const int nch = C / 8;
  const int lpr = nch <= 16 ? 16 : (nch <= 32 ? 32 : 64);
  // small inputs (BERT / GCP: 2048 rows): one pass per block so that the launch still covers the chip
  const int rpb = (yt || rows >= 64 * 2048) ? 64 : 256 / lpr;
  const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
  const size_t smem = yt ? (size_t)64 * (C + 8) * sizeof(half_t) : 0;
  if (yt && rows_per_batch <= 0) return -2;
#define MQ_LN(L)                                                                                                          \
  hipLaunchKernelGGL((layernorm_kernel<L>), dim3(grid), dim3(256), smem, (hipStream_t)stream, (const half_t*)x,          \
                     (const half_t*)res, (const half_t*)gamma, (const half_t*)beta, (half_t*)y, (half_t*)xsum, (half_t*)yt, rows, C, \
                     eps, rows_per_batch, yt_ld, rpb)
  if (nch <= 16) { MQ_LN(16); }
  else if (nch <= 32) { MQ_LN(32); }
  else { MQ_LN(64); }
#undef MQ_LN
  MQ_CHECK_LAUNCH();
  return 0;
}
