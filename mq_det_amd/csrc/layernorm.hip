// mq_layernorm_fwd: row LayerNorm for gfx950 with the residual add fused in and mixed-precision streams.
//
//   s[r, :] = x[r, :] (+ res[r, :])            x, res: fp16 or fp32 (independently)
//   y[r, :] = fp16( (s - mean) * rstd * gamma + beta )      -- the GEMM operand of whatever follows
//   y32     = the same in fp32 (optional)      -- post-LN residual stream of the BERT layers
//   xsum    = s (optional)                     -- pre-LN residual stream of the Swin / GCP blocks; fp32 when x or res
//                                                 is fp32, otherwise fp16 (s is then rounded to fp16 BEFORE the
//                                                 statistics, exactly what a separate elementwise add would produce)
//
// Every LayerNorm of the MQ-GLIP forward goes through here: Swin norm1 / norm2 / patch-merging / out norms
// (backbone/swint.py:198,240,281,611), HF BertLayer / embeddings, GCP norms (language_backbone/modeling_bert_new.py:121,
// 150-153), VLFuse norms (utils/fuse_helper.py:420-421).  The residual streams of the transformer stacks are kept in
// fp32 (round 2: fp16 re-rounding of the stream at every block was the largest avoidable term of the end-to-end error,
// DESIGN.md section 7); the normalised activations that feed MFMA GEMMs are fp16.
// A row is owned by LPR = 16 / 32 / 64 lanes (C/8 eight-element chunks, several rows per wave for small C): every access
// is a coalesced 16- or 32-byte vector; torch's LayerNorm ran the C = 96 / 192 rows of Swin stage 1-2 at ~0.4 TB/s.
#include "common.h"

MQ_NAMESPACE_BEGIN

template <bool F32>
__device__ __forceinline__ void load8(const void* base, long off, float* v) {
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

__device__ __forceinline__ void store8f(float* base, long off, const float* v) {
  float4_ a, b;
#pragma unroll
  for (int j = 0; j < 4; ++j) { a[j] = v[j]; b[j] = v[4 + j]; }
  *(float4_*)(base + off) = a;
  *(float4_*)(base + off + 4) = b;
}

template <int LPR, bool XF32, bool RF32>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ x, const void* __restrict__ res,
                                                        const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                                                        half_t* __restrict__ y, float* __restrict__ y32, void* __restrict__ xsum,
                                                        long rows, int C, float eps, int RPB, float clamp) {
  // clamp > 0 (mq_layernorm_clamp_fwd: the VLDyHead BERT copies, rpn/modeling_bert.py:242-272): x is clamped to +-clamp before the
  // residual add and both outputs after the affine -- the three torch.clamp passes around this LayerNorm, inside it
  constexpr int ROWS_PER_PASS = 256 / LPR;          // RPB = rows per block: 64 (big inputs) or one pass
  constexpr bool SUM32 = XF32 || RF32;
  const int sub = threadIdx.x % LPR, rg = threadIdx.x / LPR;
  const int nch = C / 8;
  const long r0 = (long)blockIdx.x * RPB;
  constexpr int MAXC = LPR == 64 ? 6 : 4;          // chunks per lane (C <= 8 * LPR * MAXC: 3072 for the Swin-L patch merging)
  for (int rr = rg; rr < RPB; rr += ROWS_PER_PASS) {
    const long row = r0 + rr;
    const bool ok = row < rows;
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int ch = sub + k * LPR;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[k][j] = 0.f;
      if (ok && ch < nch) {
        load8<XF32>(x, row * C + ch * 8, v[k]);
        if (clamp > 0.f) {                                  // a 16-bit x is clamped at the 16-bit value of the bound, as torch.clamp on it does
          const float cx = XF32 ? clamp : (float)(half_t)clamp;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[k][j] = __builtin_amdgcn_fmed3f(v[k][j], -cx, cx);
        }
        if (res) {
          float r[8];
          load8<RF32>(res, row * C + ch * 8, r);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[k][j] += r[j];
          if constexpr (!SUM32) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[k][j] = (float)(half_t)v[k][j];
          }
          if (xsum) {
            if constexpr (SUM32) {
              store8f((float*)xsum, row * C + ch * 8, v[k]);
            } else {
              half8 o;
#pragma unroll
              for (int j = 0; j < 8; ++j) o[j] = (half_t)v[k][j];
              *(half8*)((half_t*)xsum + row * C + ch * 8) = o;
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[k][j];
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
        for (int j = 0; j < 8; ++j) { float d = v[k][j] - mean; q += d * d; }
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int ch = sub + k * LPR;
      if (ok && ch < nch) {
        const half8 g = *(const half8*)(gamma + ch * 8), bb = *(const half8*)(beta + ch * 8);
        half8 o;
        float of[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          of[j] = (v[k][j] - mean) * rstd * (float)g[j] + (float)bb[j];
          o[j] = (half_t)of[j];
        }
        if (clamp > 0.f) {                                  // the 16-bit output: clamp of the ROUNDED value at the rounded bound, as y16.clamp() does
          const float ch_ = (float)(half_t)clamp;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            o[j] = (half_t)__builtin_amdgcn_fmed3f((float)o[j], -ch_, ch_);
            of[j] = __builtin_amdgcn_fmed3f(of[j], -clamp, clamp);
          }
        }
        if (y) *(half8*)(y + row * C + ch * 8) = o;
        if (y32) store8f(y32, row * C + ch * 8, of);
      }
    }
  }
}

static int layernorm_launch(const void* x, int x_f32, const void* res, int res_f32, const void* gamma, const void* beta,
                            void* y, float* y32, void* xsum, long rows, int C, float eps, float clamp, void* stream) {
  if (rows <= 0) return 0;
  if (C % 8 || C > 3072) return -1;
  if (!res && xsum) return -2;
  const int nch = C / 8;
  const int lpr = nch <= 16 ? 16 : (nch <= 32 ? 32 : 64);
  // small inputs (BERT / GCP: 2048 rows): one pass per block so that the launch still covers the chip
  const int rpb = rows >= 64 * 2048 ? 64 : 256 / lpr;
  const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
  const bool xf = x_f32 != 0, rf = res && res_f32 != 0;
#define MQ_LN3(L, XF, RF)                                                                                               \
  hipLaunchKernelGGL((layernorm_kernel<L, XF, RF>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, res,              \
                     (const half_t*)gamma, (const half_t*)beta, (half_t*)y, y32, xsum, rows, C, eps, rpb, clamp)
#define MQ_LN(L)                                                                                                        \
  do {                                                                                                                  \
    if (xf && rf) MQ_LN3(L, true, true);                                                                                \
    else if (xf) MQ_LN3(L, true, false);                                                                                \
    else if (rf) MQ_LN3(L, false, true);                                                                                \
    else MQ_LN3(L, false, false);                                                                                       \
  } while (0)
  if (nch <= 16) MQ_LN(16);
  else if (nch <= 32) MQ_LN(32);
  else MQ_LN(64);
#undef MQ_LN
#undef MQ_LN3
  MQ_CHECK_LAUNCH();
  return 0;
}

extern "C" int MQ_SYM(mq_layernorm_fwd)(const void* x, int x_f32, const void* res, int res_f32, const void* gamma, const void* beta,
                                void* y, float* y32, void* xsum, long rows, int C, float eps, void* stream) {
  return layernorm_launch(x, x_f32, res, res_f32, gamma, beta, y, y32, xsum, rows, C, eps, 0.f, stream);
}

// LayerNorm(clamp(x) (+ res)) with both outputs clamped: y = clamp(round16(LN)), y32 = clamp(LN) -- equal to
// x.clamp(-c, c) -> mq_layernorm_fwd -> y.clamp(-c, c), y32.clamp(-c, c) bit for bit (same kernel, same order), three passes less.
extern "C" int MQ_SYM(mq_layernorm_clamp_fwd)(const void* x, int x_f32, const void* res, int res_f32, const void* gamma, const void* beta,
                                      void* y, float* y32, void* xsum, long rows, int C, float eps, float clamp, void* stream) {
  if (!(clamp > 0.f)) return -1;
  return layernorm_launch(x, x_f32, res, res_f32, gamma, beta, y, y32, xsum, rows, C, eps, clamp, stream);
}

// out = clamp(gelu(clamp(x))) elementwise, 16-bit in / out, exact (erf) GELU evaluated in fp32 and rounded once like torch's kernel:
// F.gelu(h.clamp(-c, c)).clamp(-c, c) of the clamped BERT copies (rpn/modeling_bert.py:255-259) in one pass instead of three.
__global__ __launch_bounds__(256) void clamp_gelu_clamp_kernel(const half_t* __restrict__ x, half_t* __restrict__ out, long n8, float clamp) {
  clamp = (float)(half_t)clamp;                              // torch.clamp on a 16-bit tensor compares with the 16-bit value of the bound
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const half8 v = *(const half8*)(x + i * 8);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = __builtin_amdgcn_fmed3f((float)v[j], -clamp, clamp);
      const float gl = a * 0.5f * (1.f + erff(a * 0.70710678118654752440f));
      o[j] = (half_t)__builtin_amdgcn_fmed3f((float)(half_t)gl, -clamp, clamp);
    }
    *(half8*)(out + i * 8) = o;
  }
}

extern "C" int MQ_SYM(mq_clamp_gelu_clamp)(const void* x, void* out, long n, float clamp, void* stream) {
  if (n <= 0) return 0;
  if (n % 8 || !(clamp > 0.f)) return -1;
  const long n8 = n / 8;
  const long blocks = (n8 + 255) / 256;
  hipLaunchKernelGGL(clamp_gelu_clamp_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)x, (half_t*)out, n8, clamp);
  MQ_CHECK_LAUNCH();
  return 0;
}

MQ_NAMESPACE_END
