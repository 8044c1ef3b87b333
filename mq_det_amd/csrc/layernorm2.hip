// mq_layernorm2_fwd: the same operator as mq_layernorm_fwd (layernorm.hip: row LayerNorm with the residual add fused in, fp16 / fp32
// streams), same arguments, bit-identical results (the summation order is kept) -- a different load schedule.
//
// tools/isa_wait_scan.py on layernorm.hip: 4 - 12 of the 16 - 36 global loads of a row iteration are followed by `s_waitcnt vmcnt(0)`
// at once.  The chunk loop there always spans 4 (6) chunks with `ch < nch` guards although C <= 512 needs ONE chunk per lane, x and
// residual are loaded and consumed chunk by chunk, and gamma / beta are fetched again for every row right before they are used: a
// row costs several dependent memory round trips, which is what a 2.8 TB/s (35 % of HBM) LayerNorm looks like.
// Here: the chunk count per lane is a template parameter (1, 2, 3, 4, 6), gamma / beta live in registers for the whole block, and
// R rows per lane group are in flight together (R = 4 for one chunk per lane, 2 for two, 1 above): every load of an iteration is
// issued before the first one is consumed.
// Default since round 3 (KERNELS["LN_VARIANT"] = 2: 1.39 -> 1.13 ms of LayerNorm per step, 2.8 -> 3.0 TB/s); bit for bit the results of
// mq_layernorm_fwd on the device and through tests/simt.
#include "common.h"

MQ_NAMESPACE_BEGIN

namespace {
template <bool F32>
__device__ __forceinline__ void ln2_load8(const void* base, long off, float* v) {
  if constexpr (F32) {
    const float4_ a = *(const float4_*)((const float*)base + off), b = *(const float4_*)((const float*)base + off + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
  } else {
    const half8 a = *(const half8*)((const half_t*)base + off);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
  }
}
__device__ __forceinline__ void ln2_store8f(float* base, long off, const float* v) {
  float4_ a, b;
#pragma unroll
  for (int j = 0; j < 4; ++j) { a[j] = v[j]; b[j] = v[4 + j]; }
  *(float4_*)(base + off) = a;
  *(float4_*)(base + off + 4) = b;
}
}  // namespace

template <int LPR, int MAXC, int R, bool XF32, bool RF32, bool HAS_RES>
__global__ __launch_bounds__(256) void layernorm2_kernel(const void* __restrict__ x, const void* __restrict__ res,
                                                         const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                                                         half_t* __restrict__ y, float* __restrict__ y32, void* __restrict__ xsum,
                                                         long rows, int C, float eps, int RPB) {
  constexpr int GROUPS = 256 / LPR;                 // lane groups (rows) per pass of a workgroup
  constexpr bool SUM32 = XF32 || RF32;
  const int sub = threadIdx.x % LPR, rg = threadIdx.x / LPR;
  const int nch = C / 8;
  const long r0 = (long)blockIdx.x * RPB;
  // chunk k of this lane: clamped to a valid chunk for the loads (no guards around them), `live` decides what is used / stored
  int chk[MAXC];
  bool live[MAXC];
  half8 g[MAXC], bt[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int ch = sub + k * LPR;
    live[k] = ch < nch;
    chk[k] = live[k] ? ch : nch - 1;
    g[k] = *(const half8*)(gamma + chk[k] * 8);
    bt[k] = *(const half8*)(beta + chk[k] * 8);
  }
  for (int rr = rg * R; rr < RPB; rr += GROUPS * R) {
    float v[R][MAXC][8];
    float rv[HAS_RES ? R : 1][MAXC][8];
    bool ok[R];
    long rowc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const long row = r0 + rr + i;
      ok[i] = rr + i < RPB && row < rows;
      rowc[i] = min(row, rows - 1);                  // out-of-range rows re-read the last row and store nothing
    }
    // ---- every load of this iteration, then the arithmetic
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        ln2_load8<XF32>(x, rowc[i] * C + chk[k] * 8, v[i][k]);
        if constexpr (HAS_RES) ln2_load8<RF32>(res, rowc[i] * C + chk[k] * 8, rv[i][k]);
      }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        if constexpr (HAS_RES) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[i][k][j] += rv[i][k][j];
          if constexpr (!SUM32) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][k][j] = (float)(half_t)v[i][k][j];
          }
          if (xsum && ok[i] && live[k]) {
            if constexpr (SUM32) {
              ln2_store8f((float*)xsum, rowc[i] * C + chk[k] * 8, v[i][k]);
            } else {
              half8 o;
#pragma unroll
              for (int j = 0; j < 8; ++j) o[j] = (half_t)v[i][k][j];
              *(half8*)((half_t*)xsum + rowc[i] * C + chk[k] * 8) = o;
            }
          }
        }
        if (!live[k]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[i][k][j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[i][k][j];
      }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
      const float mean = s / (float)C;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        if (live[k]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { float d = v[i][k][j] - mean; q += d * d; }
        }
      }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
      const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        if (ok[i] && live[k]) {
          half8 o;
          float of[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            of[j] = (v[i][k][j] - mean) * rstd * (float)g[k][j] + (float)bt[k][j];
            o[j] = (half_t)of[j];
          }
          if (y) *(half8*)(y + rowc[i] * C + chk[k] * 8) = o;
          if (y32) ln2_store8f(y32, rowc[i] * C + chk[k] * 8, of);
        }
      }
    }
  }
}

template <int LPR, int MAXC, int R>
static void launch_ln2(const void* x, bool xf, const void* res, bool rf, const void* gamma, const void* beta, void* y, float* y32, void* xsum,
                       long rows, int C, float eps, hipStream_t stream) {
  constexpr int per_pass = (256 / LPR) * R;
  // small inputs (BERT / GCP: 2048 rows): one pass per block so that the launch still covers the chip
  const int rpb = rows >= 64 * 2048 ? (64 > per_pass ? 64 : per_pass) : per_pass;
  const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
#define MQ_LN2(XF, RF, HR)                                                                                                     \
  hipLaunchKernelGGL((layernorm2_kernel<LPR, MAXC, R, XF, RF, HR>), dim3(grid), dim3(256), 0, stream, x, res, (const half_t*)gamma, \
                     (const half_t*)beta, (half_t*)y, y32, xsum, rows, C, eps, rpb)
  if (!res) { if (xf) MQ_LN2(true, false, false); else MQ_LN2(false, false, false); }
  else if (xf && rf) MQ_LN2(true, true, true);
  else if (xf) MQ_LN2(true, false, true);
  else if (rf) MQ_LN2(false, true, true);
  else MQ_LN2(false, false, true);
#undef MQ_LN2
}

extern "C" int MQ_SYM(mq_layernorm2_fwd)(const void* x, int x_f32, const void* res, int res_f32, const void* gamma, const void* beta,
                                         void* y, float* y32, void* xsum, long rows, int C, float eps, void* stream) {
  if (rows <= 0) return 0;
  if (C % 8 || C > 3072) return -1;
  if (!res && xsum) return -2;
  const int nch = C / 8;
  const bool xf = x_f32 != 0, rf = res && res_f32 != 0;
  hipStream_t s = (hipStream_t)stream;
  if (nch <= 16) launch_ln2<16, 1, 4>(x, xf, res, rf, gamma, beta, y, y32, xsum, rows, C, eps, s);
  else if (nch <= 32) launch_ln2<32, 1, 4>(x, xf, res, rf, gamma, beta, y, y32, xsum, rows, C, eps, s);
  else if (nch <= 64) launch_ln2<64, 1, 4>(x, xf, res, rf, gamma, beta, y, y32, xsum, rows, C, eps, s);
  else if (nch <= 128) launch_ln2<64, 2, 2>(x, xf, res, rf, gamma, beta, y, y32, xsum, rows, C, eps, s);
  else if (nch <= 192) launch_ln2<64, 3, 1>(x, xf, res, rf, gamma, beta, y, y32, xsum, rows, C, eps, s);
  else if (nch <= 256) launch_ln2<64, 4, 1>(x, xf, res, rf, gamma, beta, y, y32, xsum, rows, C, eps, s);
  else launch_ln2<64, 6, 1>(x, xf, res, rf, gamma, beta, y, y32, xsum, rows, C, eps, s);
  MQ_CHECK_LAUNCH();
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------------------
// mq_patch_merge_ln_fwd: Swin PatchMerging up to its LayerNorm (backbone/swint.py:258-284) in one kernel:
//   y[b, i, j, :] = LayerNorm_{4C}( concat( x[b, 2i, 2j], x[b, 2i+1, 2j], x[b, 2i, 2j+1], x[b, 2i+1, 2j+1] ) )   (zero beyond an odd H / W)
// The host path does F.pad + four strided slices + torch.cat (a 0.24 ms / step CatArrayBatchedCopy pass that writes the gathered
// [B, H/2, W/2, 4C] fp32 tensor, profiles/r02_call5) and then mq_layernorm_fwd reads it back; here the LayerNorm reads the four
// C-wide segments of a row straight from x.  Same lane <-> chunk assignment and summation order as mq_layernorm_fwd on the
// concatenated row: the result equals cat + LayerNorm bit for bit.  x: [B, H, W, C] fp16 or fp32 (x_f32), contiguous; y: fp16
// [B, ceil(H/2) * ceil(W/2), 4C]; C % 8 == 0, 4C <= 3072.  Default since round 3 (KERNELS["PATCH_MERGE_FUSED"] = 1).
template <int LPR, int MAXC, int R, bool XF32>
__global__ __launch_bounds__(256) void patch_merge_ln_kernel(const void* __restrict__ x, const half_t* __restrict__ gamma,
                                                             const half_t* __restrict__ beta, half_t* __restrict__ y, int B, int H, int W,
                                                             int C, float eps, int RPB) {
  constexpr int GROUPS = 256 / LPR;
  const int sub = threadIdx.x % LPR, rg = threadIdx.x / LPR;
  const int C4 = 4 * C, nch = C4 / 8;
  const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
  const long rows = (long)B * H2 * W2;
  const long r0 = (long)blockIdx.x * RPB;
  int chk[MAXC], seg_dy[MAXC], seg_dx[MAXC], seg_off[MAXC];
  bool live[MAXC];
  half8 g[MAXC], bt[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int ch = sub + k * LPR;
    live[k] = ch < nch;
    chk[k] = live[k] ? ch : nch - 1;
    const int e = chk[k] * 8, seg = e / C;                 // a chunk never straddles two segments (C % 8 == 0)
    seg_dy[k] = seg & 1;                                   // concat order x0 (0,0), x1 (1,0), x2 (0,1), x3 (1,1): (dy, dx)
    seg_dx[k] = seg >> 1;
    seg_off[k] = e - seg * C;
    g[k] = *(const half8*)(gamma + chk[k] * 8);
    bt[k] = *(const half8*)(beta + chk[k] * 8);
  }
  for (int rr = rg * R; rr < RPB; rr += GROUPS * R) {
    float v[R][MAXC][8];
    bool ok[R], in[R][MAXC];
    long rowc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const long row = r0 + rr + i;
      ok[i] = rr + i < RPB && row < rows;
      rowc[i] = min(row, rows - 1);
      const int b = (int)(rowc[i] / ((long)H2 * W2)), rem = (int)(rowc[i] % ((long)H2 * W2));
      const int hi = rem / W2, wi = rem % W2;
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        const int hh = 2 * hi + seg_dy[k], ww = 2 * wi + seg_dx[k];
        in[i][k] = hh < H && ww < W;                       // beyond an odd H / W: the zero padding of swint.py:270-272
        const long pix = ((long)b * H + min(hh, H - 1)) * W + min(ww, W - 1);
        ln2_load8<XF32>(x, pix * C + seg_off[k], v[i][k]);
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        if (!live[k] || !in[i][k]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[i][k][j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[i][k][j];
      }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
      const float mean = s / (float)C4;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        if (live[k]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { float d = v[i][k][j] - mean; q += d * d; }
        }
      }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
      const float rstd = rsqrtf(q / (float)C4 + eps);
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        if (ok[i] && live[k]) {
          half8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (half_t)((v[i][k][j] - mean) * rstd * (float)g[k][j] + (float)bt[k][j]);
          *(half8*)(y + rowc[i] * C4 + chk[k] * 8) = o;
        }
      }
    }
  }
}

template <int LPR, int MAXC, int R>
static void launch_pm(const void* x, bool xf, const void* gamma, const void* beta, void* y, int B, int H, int W, int C, float eps, hipStream_t stream) {
  constexpr int per_pass = (256 / LPR) * R;
  const long rows = (long)B * ((H + 1) / 2) * ((W + 1) / 2);
  const int rpb = rows >= 64 * 2048 ? (64 > per_pass ? 64 : per_pass) : per_pass;
  const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
  if (xf) hipLaunchKernelGGL((patch_merge_ln_kernel<LPR, MAXC, R, true>), dim3(grid), dim3(256), 0, stream, x, (const half_t*)gamma,
                             (const half_t*)beta, (half_t*)y, B, H, W, C, eps, rpb);
  else hipLaunchKernelGGL((patch_merge_ln_kernel<LPR, MAXC, R, false>), dim3(grid), dim3(256), 0, stream, x, (const half_t*)gamma,
                          (const half_t*)beta, (half_t*)y, B, H, W, C, eps, rpb);
}

extern "C" int MQ_SYM(mq_patch_merge_ln_fwd)(const void* x, int x_f32, const void* gamma, const void* beta, void* y, int B, int H, int W, int C,
                                             float eps, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  if (C % 8 || 4 * C > 3072) return -1;
  const int nch = 4 * C / 8;
  const bool xf = x_f32 != 0;
  hipStream_t s = (hipStream_t)stream;
  if (nch <= 16) launch_pm<16, 1, 4>(x, xf, gamma, beta, y, B, H, W, C, eps, s);
  else if (nch <= 32) launch_pm<32, 1, 4>(x, xf, gamma, beta, y, B, H, W, C, eps, s);
  else if (nch <= 64) launch_pm<64, 1, 4>(x, xf, gamma, beta, y, B, H, W, C, eps, s);
  else if (nch <= 128) launch_pm<64, 2, 2>(x, xf, gamma, beta, y, B, H, W, C, eps, s);
  else if (nch <= 192) launch_pm<64, 3, 1>(x, xf, gamma, beta, y, B, H, W, C, eps, s);
  else if (nch <= 256) launch_pm<64, 4, 1>(x, xf, gamma, beta, y, B, H, W, C, eps, s);
  else launch_pm<64, 6, 1>(x, xf, gamma, beta, y, B, H, W, C, eps, s);
  MQ_CHECK_LAUNCH();
  return 0;
}

// ---- mq_dyrelu_ln_fwd: y = LayerNorm(DYReLU(x)) on the pyramid token buffer of the head.
// The only reader of a DyConv layer's output is layer_norm_v of the NEXT fusion layer (fuse_helper.py:398: both VLFuse directions and
// the residual take the NORMED v), so DYReLU -- max(a1 x + b1, a2 x + b2) with per-(image, LEVEL, channel) coefficients
// (vldyhead.py:160-188, 245) -- need not be a pass of its own: 25 of the 30 mq_dyrelu_apply launches of a forward and their
// read + write of the buffer go.  The DYReLU result is consumed in fp32 (the stand-alone kernel rounds it to 16 bits first).
// x, y [B, N, 256] 16-bit; coef [NL][B][4][256] fp32 (a1, b1, a2, b2 -- mq_dyrelu_coef's layout per level); level l = rows
// [row_first[l], row_first[l + 1]) of every image.  A workgroup = 32 rows of ONE level (its coefficients live in registers): 8 row
// groups of 32 lanes x 8 channels, 4 rows per group, all loads in flight before the first is consumed.
struct DyreluLnParams {
  const half_t* x; const float* coef; const half_t* gamma; const half_t* beta; half_t* y;
  long x_bs;
  int B, N, NL;
  int row_first[9], blk_first[9];
  float eps;
};

__global__ __launch_bounds__(256) void dyrelu_ln_kernel(DyreluLnParams p) {
  constexpr int C = 256, RPB = 32;
  const int b = blockIdx.y;
  int l = 0;
  while (l + 1 < p.NL && (int)blockIdx.x >= p.blk_first[l + 1]) ++l;
  const int sub = threadIdx.x & 31, grp = threadIdx.x >> 5, c0 = sub * 8;
  const int row_end = p.row_first[l + 1];
  const int row0 = p.row_first[l] + ((int)blockIdx.x - p.blk_first[l]) * RPB + grp;
  const half_t* xb = p.x + (long)b * p.x_bs;
  half8 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = *(const half8*)(xb + (long)min(row0 + 8 * k, row_end - 1) * C + c0);
  const float* cf = p.coef + ((long)l * p.B + b) * 4 * C + c0;
  float a1[8], b1[8], a2[8], b2[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float4_ t0 = *(const float4_*)(cf + 4 * h), t1 = *(const float4_*)(cf + C + 4 * h);
    const float4_ t2 = *(const float4_*)(cf + 2 * C + 4 * h), t3 = *(const float4_*)(cf + 3 * C + 4 * h);
#pragma unroll
    for (int j = 0; j < 4; ++j) { a1[4 * h + j] = t0[j]; b1[4 * h + j] = t1[j]; a2[4 * h + j] = t2[j]; b2[4 * h + j] = t3[j]; }
  }
  const half8 g = *(const half8*)(p.gamma + c0), bt = *(const half8*)(p.beta + c0);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float f[8], s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (float)v[k][j];
      f[j] = fmaxf(x * a1[j] + b1[j], x * a2[j] + b2[j]);
      s += f[j];
    }
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) s += __shfl_xor(s, m);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = f[j] - mean; q += d * d; }
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) q += __shfl_xor(q, m);
    const float rstd = rsqrtf(q / (float)C + p.eps);
    const int row = row0 + 8 * k;
    if (row < row_end) {
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)((f[j] - mean) * rstd * (float)g[j] + (float)bt[j]);
      *(half8*)(p.y + ((long)b * p.N + row) * C + c0) = o;
    }
  }
}

// row_first: NL + 1 host ints (row_first[0] = 0, row_first[NL] = N).  C must be 256, NL <= 8.
extern "C" int MQ_SYM(mq_dyrelu_ln_fwd)(const void* x, long x_bs, const float* coef, const int* row_first, int NL, const void* gamma,
                                        const void* beta, float eps, void* y, int B, int N, int C, void* stream) {
  if (B <= 0 || N <= 0) return 0;
  if (C != 256 || NL < 1 || NL > 8 || row_first[0] != 0 || row_first[NL] != N) return -1;
  DyreluLnParams p;
  p.x = (const half_t*)x; p.coef = coef; p.gamma = (const half_t*)gamma; p.beta = (const half_t*)beta; p.y = (half_t*)y;
  p.x_bs = x_bs; p.B = B; p.N = N; p.NL = NL; p.eps = eps;
  int blk = 0;
  for (int l = 0; l <= NL; ++l) {
    p.row_first[l] = row_first[l];
    p.blk_first[l] = blk;
    if (l < NL) {
      if (row_first[l + 1] <= row_first[l]) return -1;
      blk += (row_first[l + 1] - row_first[l] + 31) / 32;
    }
  }
  hipLaunchKernelGGL(dyrelu_ln_kernel, dim3((unsigned)blk, (unsigned)B), dim3(256), 0, (hipStream_t)stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

MQ_NAMESPACE_END
