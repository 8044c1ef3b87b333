// mq_conv3x3_fwd: 3x3 convolution as ONE implicit-GEMM MFMA kernel, gfx950.  (The DEFORM = true instantiation is the
// 4-wave v1 of the fused DCNv2; the shipped mq_dcnv2_fwd is the 8-wave kernel in dcn_fused.hip.)
//
//   out[m, n] = bias[n] + sum_{tap, c} A[m, tap, c] * W[n, tap*C + c],      m = (b, ho, wo), NHWC fp16 in / out
//   plain  : A[m, tap, c] = x[b, ho*s - 1 + ky, wo*s - 1 + kx, c]            (zero padding)
//   deform : A[m, tap, c] = sigmoid(mask_logit) * bilinear(x[b, :, :, c], ho*s - 1 + ky + dh, wo*s - 1 + kx + dw)
//
// Reference: DCNv2 forward = maskrcnn_benchmark/csrc/cuda/deform_conv_kernel_cuda.cu:578-640 (im2col, bilinear
// :475-503) + deform_conv_cuda.cu:538-560 (per-sample fp32 column buffer [C*9, Ho*Wo] -> addmm).  The column matrix
// (155 MB fp32 per image for P3 in the reference, 1.24 GB fp16 per DyConv layer in v1 of this repo) never exists
// here: the bilinear gather writes MFMA A-tiles straight into LDS.  Quirk kept: offsets / mask
// logits come from om[b, 27, oH, oW] (fp32, NCHW) and are indexed FLAT by the OUTPUT dims, so the buffer may come from
// a different pyramid level.  The plain variant replaces MIOpen for the FPN 3x3 convs and the 27-channel offset conv
// (MIOpen's fp16 solvers there are not run-to-run reproducible; this kernel accumulates the whole K = 9*C in fp32 in
// a fixed order and is).
//
// Tiling: 128 output positions x BN output channels x 32 k per step, 4 waves (WM x WN), v_mfma_f32_16x16x32_f16,
// double-buffered LDS tiles (row pitch 32+8 halfs), global loads of step k+1 in flight during the MFMAs of step k,
// one barrier per step; epilogue transposes through LDS for 16-byte coalesced NHWC stores.
#include "common.h"

MQ_NAMESPACE_BEGIN

struct ConvParams {
  const half_t* x;       // [B, H, W, C], batch stride x_bs (elements)
  const half_t* w;       // [BN rows][9*C]  (k = tap*C + c), rows >= N are zero
  const half_t* bias;    // [N] or nullptr
  const float* om;       // deform: [B, 27, oH, oW]
  half_t* out;           // [B*Ho*Wo, out_ld]
  long x_bs;
  int B, H, W, C, Ho, Wo, stride, N, out_ld, oH, oW;
};

template <bool DEFORM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvParams p) {
  constexpr int BM = 128, BK = 32, LP = BK + 8;          // LDS row pitch (halfs)
  constexpr int RBW = BM / WM / 16, CBW = BN / WN / 16;   // 16x16 blocks per wave
  constexpr int NBCH = (BN * 4 + 255) / 256;              // B-tile 16-byte chunks per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* As = (half_t*)smem;                             // [2][BM][LP]
  half_t* Bs = As + 2 * BM * LP;                          // [2][BN][LP]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int wr = wave / WN, wc = wave % WN;
  const int n_pos = p.Ho * p.Wo;
  const long M = (long)p.B * n_pos;
  // XCD-aware tile order (workgroup i runs on XCD i % 8, one 4 MB L2 per XCD): XCD x owns the contiguous tile range
  // [x * tpx, (x + 1) * tpx), so the 3-row halo shared by vertically adjacent tiles is re-used from ITS L2.
  const long ntiles = (M + BM - 1) / BM, tpx = (ntiles + 7) >> 3;
  const long tile = (blockIdx.x & 7) * tpx + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const long m0 = tile * BM;
  const int K = 9 * p.C;
  const int ksteps = K / BK;
  const int steps_per_tap = p.C / BK;

  // ---- the two A rows this thread produces (row = tid/4 and 64 + tid/4, channel chunk = tid%4)
  const int ach = tid & 3;
  int r_b[2], r_ho[2], r_wo[2], r_pos[2];
  bool r_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    long m = m0 + (tid >> 2) + i * 64;
    r_ok[i] = m < M;
    long mc = r_ok[i] ? m : M - 1;
    r_b[i] = (int)(mc / n_pos);
    r_pos[i] = (int)(mc % n_pos);
    r_ho[i] = r_pos[i] / p.Wo;
    r_wo[i] = r_pos[i] % p.Wo;
  }
  // per-tap sampling state
  int c_off[2][DEFORM ? 4 : 1];       // element offset of each corner inside its image (or -1)
  long r_base[2];
  float c_w[2][DEFORM ? 4 : 1];       // corner weight * mask (deform)
  auto setup_tap = [&](int tap) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int hb = r_ho[i] * p.stride - 1 + tap / 3, wb = r_wo[i] * p.stride - 1 + tap % 3;
      r_base[i] = (long)r_b[i] * p.x_bs;
      if constexpr (DEFORM) {
        const float* omb = p.om + (long)r_b[i] * 27 * p.oH * p.oW;
        const float dh = omb[(long)(2 * tap) * n_pos + r_pos[i]];
        const float dw = omb[(long)(2 * tap + 1) * n_pos + r_pos[i]];
        const float ml = omb[(long)18 * p.oH * p.oW + (long)tap * n_pos + r_pos[i]];
        const float mk = 1.f / (1.f + __expf(-ml));
        const float hf = (float)hb + dh, wf = (float)wb + dw;
        const bool inside = r_ok[i] && hf > -1.f && wf > -1.f && hf < (float)p.H && wf < (float)p.W;
        const int h0 = (int)floorf(hf), w0 = (int)floorf(wf);
        const float lh = hf - (float)h0, lw = wf - (float)w0;
        const float wq[4] = {(1.f - lh) * (1.f - lw), (1.f - lh) * lw, lh * (1.f - lw), lh * lw};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int hh = h0 + (q >> 1), ww = w0 + (q & 1);
          const bool ok = inside && hh >= 0 && hh <= p.H - 1 && ww >= 0 && ww <= p.W - 1;
          c_off[i][q] = ok ? (hh * p.W + ww) * p.C : -1;
          c_w[i][q] = ok ? wq[q] * mk : 0.f;
        }
      } else {
        const bool ok = r_ok[i] && hb >= 0 && hb < p.H && wb >= 0 && wb < p.W;
        c_off[i][0] = ok ? (hb * p.W + wb) * p.C : -1;
        c_w[i][0] = 1.f;
      }
    }
  };

  constexpr int NC = DEFORM ? 4 : 1;
  half8 a_raw[2][NC];
  half8 b_raw[NBCH];
  auto issue = [&](int ks) {
    const int cbase = (ks % steps_per_tap) * BK + ach * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        // invalid corners read a harmless valid address and are zero-weighted / zeroed below
        const int off = c_off[i][q] >= 0 ? c_off[i][q] : 0;
        a_raw[i][q] = *(const half8*)(p.x + r_base[i] + off + cbase);
      }
#pragma unroll
    for (int j = 0; j < NBCH; ++j) {
      const int c = tid + j * 256;
      if (BN * 4 >= 256 * (j + 1) || c < BN * 4)
        b_raw[j] = *(const half8*)(p.w + (long)(c >> 2) * K + ks * BK + (c & 3) * 8);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      half8 v;
      if constexpr (DEFORM) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += c_w[i][q] * (float)a_raw[i][q][j];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (half_t)acc[j];
      } else {
        v = c_off[i][0] >= 0 ? a_raw[i][0] : zero8();
      }
      *(half8*)(As + (buf * BM + (tid >> 2) + i * 64) * LP + ach * 8) = v;
    }
#pragma unroll
    for (int j = 0; j < NBCH; ++j) {
      const int c = tid + j * 256;
      if (BN * 4 >= 256 * (j + 1) || c < BN * 4) *(half8*)(Bs + (buf * BN + (c >> 2)) * LP + (c & 3) * 8) = b_raw[j];
    }
  };

  float4_ acc[RBW][CBW];
#pragma unroll
  for (int i = 0; i < RBW; ++i)
#pragma unroll
    for (int j = 0; j < CBW; ++j) acc[i][j] = (float4_){0.f, 0.f, 0.f, 0.f};

  setup_tap(0);
  issue(0);
  commit(0);
  __syncthreads();
  for (int ks = 0; ks < ksteps; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < ksteps) {
      if ((ks + 1) % steps_per_tap == 0) setup_tap((ks + 1) / steps_per_tap);
      issue(ks + 1);
    }
    half8 af[RBW];
#pragma unroll
    for (int i = 0; i < RBW; ++i) af[i] = *(const half8*)(As + (cur * BM + wr * (RBW * 16) + i * 16 + l15) * LP + lg * 8);
#pragma unroll
    for (int j = 0; j < CBW; ++j) {
      const half8 bf = *(const half8*)(Bs + (cur * BN + wc * (CBW * 16) + j * 16 + l15) * LP + lg * 8);
#pragma unroll
      for (int i = 0; i < RBW; ++i) acc[i][j] = mfma16(af[i], bf, acc[i][j]);
    }
    if (ks + 1 < ksteps) commit(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: + bias, fp16, transpose through LDS, 16-byte coalesced stores of the real N columns
  constexpr int OS = BN + 8;
  half_t* Os = (half_t*)smem;                              // [BM][OS]  (tiles are dead; last barrier passed)
#pragma unroll
  for (int j = 0; j < CBW; ++j) {
    const int col = wc * (CBW * 16) + j * 16 + l15;
    const float bv = (p.bias && col < p.N) ? (float)p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Os[(wr * (RBW * 16) + i * 16 + lg * 4 + r) * OS + col] = (half_t)(acc[i][j][r] + bv);
  }
  __syncthreads();
  const int chunks_per_row = (p.N + 7) / 8;
  for (int c = tid; c < BM * chunks_per_row; c += 256) {
    const int row = c / chunks_per_row, ch = c % chunks_per_row;
    const long m = m0 + row;
    if (m < M) {
      if (ch * 8 + 8 <= p.N && (p.out_ld % 8) == 0) {
        *(half8*)(p.out + m * p.out_ld + ch * 8) = *(const half8*)(Os + row * OS + ch * 8);
      } else {
        for (int j = 0; j < 8 && ch * 8 + j < p.N; ++j) p.out[m * p.out_ld + ch * 8 + j] = Os[row * OS + ch * 8 + j];
      }
    }
  }
}

template <bool DEFORM, int BN, int WM, int WN>
static int launch_conv(const ConvParams& p, hipStream_t stream) {
  constexpr int BM = 128, LP = 40;
  constexpr size_t tiles = (size_t)(2 * BM * LP + 2 * BN * LP) * sizeof(half_t);
  constexpr size_t ostage = (size_t)BM * (BN + 8) * sizeof(half_t);
  constexpr size_t smem = tiles > ostage ? tiles : ostage;
  static MqOncePerDevice attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_igemm_kernel<DEFORM, BN, WM, WN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set.done();
  }
  long M = (long)p.B * p.Ho * p.Wo;
  hipLaunchKernelGGL((conv_igemm_kernel<DEFORM, BN, WM, WN>), dim3((unsigned)(8 * (((M + BM - 1) / BM + 7) / 8))), dim3(256), smem, stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

static int conv_common(ConvParams& p, const void* x, const void* w, const void* bias, void* out, int B, int H, int W, int C,
                       long x_bs, int N, int out_ld, int stride) {
  if (B <= 0) return 0;
  if (C % 32 || stride < 1 || stride > 2 || N < 1 || out_ld < N) return -1;
  p.x = (const half_t*)x; p.w = (const half_t*)w; p.bias = (const half_t*)bias; p.out = (half_t*)out;
  p.x_bs = x_bs; p.B = B; p.H = H; p.W = W; p.C = C; p.stride = stride; p.N = N; p.out_ld = out_ld;
  p.Ho = (H + 2 - 3) / stride + 1; p.Wo = (W + 2 - 3) / stride + 1;
  p.om = nullptr; p.oH = p.oW = 0;
  return 1;
}

// plain 3x3 conv, pad 1.  w: [Npad, 9*C] with Npad = 256 (N <= 256, N > 32) or 32 (N <= 32), rows >= N zero.
extern "C" int MQ_SYM(mq_conv3x3_fwd)(const void* x, const void* w, const void* bias, void* out, int B, int H, int W, int C,
                              long x_bs, int N, int out_ld, int stride, void* stream) {
  ConvParams p;
  int rc = conv_common(p, x, w, bias, out, B, H, W, C, x_bs, N, out_ld, stride);
  if (rc <= 0) return rc;
  if (N <= 32) return launch_conv<false, 32, 4, 1>(p, (hipStream_t)stream);
  if (N <= 256) return launch_conv<false, 256, 2, 2>(p, (hipStream_t)stream);
  return -1;
}

MQ_NAMESPACE_END
