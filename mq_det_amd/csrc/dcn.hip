// mq_dcn_im2col_fwd: modulated deformable (DCNv2) 3x3 column gather for gfx950, NHWC fp16.
//
// Reference: csrc/cuda/deform_conv_kernel_cuda.cu:578-640 (im2col kernel), :475-503 (bilinear) and
// deform_conv_cuda.cu:538-560 (per-sample fp32 column buffer [C*9, Ho*Wo] + addmm).  Differences:
//   * whole batch in one launch, fp16 NHWC activations: every bilinear corner is one contiguous
//     C-vector (C = 256 -> 512 B), each lane moves 8 channels (16 B) -> fully coalesced gathers;
//   * columns are written token-major / K-contiguous  cols[b, n, tap*C + c]  so the following GEMM
//     (library, against the weight repacked to [O, tap*C + c]) yields NHWC output directly;
//   * sigmoid of the mask logits is applied here (reference: separate elementwise op, vldyhead.py:216).
// Quirk kept: offsets / mask logits come from ONE tensor om[b, 27, oH, oW]
//   (channels 0..17 = (dh, dw) per tap, 18..26 = mask logits) that may have been computed on a
//   DIFFERENT pyramid level than the conv's output: the reference kernel indexes the per-sample offset
//   buffer flat with the OUTPUT dims -- element ((2k)*Ho + h)*Wo + w of the [18, oH, oW] block (and
//   (k*Ho + h)*Wo + w of the [9, oH, oW] mask block) -- which is what is reproduced here.
#include "common.h"

struct DcnParams {
  const half_t* x;      // [B, H, W, C]
  const float* om;      // [B, 27, oH, oW] fp32, NCHW contiguous
  half_t* cols;         // [B, Ho*Wo, 9*C]
  int B, H, W, C, oH, oW, Ho, Wo, stride;
};

__global__ __launch_bounds__(256) void dcn_im2col_kernel(DcnParams p) {
  const int cpt = p.C / 8;                       // lanes per (position, tap)
  const int pairs_per_block = 256 / cpt;
  const int sub = threadIdx.x / cpt;             // which (position, tap) pair inside the block
  const int c0 = (threadIdx.x % cpt) * 8;
  const long npos = (long)p.B * p.Ho * p.Wo;
  const long total = npos * 9;
  const int n = p.Ho * p.Wo;
  const long om_plane = (long)p.oH * p.oW;
  // XCD-aware order (workgroup i runs on XCD i % 8, one L2 per XCD): XCD x sweeps the contiguous pair range
  // [x * chunk, (x + 1) * chunk) so that the re-use of input rows between neighbouring positions / taps hits ITS L2.
  const long chunk = ((total + 8L * pairs_per_block - 1) / (8L * pairs_per_block)) * pairs_per_block;
  const int xcd = blockIdx.x & 7, lblk = blockIdx.x >> 3, nlblk = gridDim.x >> 3;
  const long pend = min(total, (xcd + 1) * chunk);
  for (long pair = xcd * chunk + (long)lblk * pairs_per_block + sub; pair < pend; pair += (long)nlblk * pairs_per_block) {
    const int k = pair % 9;
    const long pos = pair / 9;
    const int b = pos / n, rem = pos % n;
    const int ho = rem / p.Wo, wo = rem % p.Wo;
    const float* omb = p.om + (long)b * 27 * om_plane;
    // flat indexing by OUTPUT dims into the [18, oH, oW] / [9, oH, oW] blocks
    const float dh = omb[(long)(2 * k) * n + rem];
    const float dw = omb[(long)(2 * k + 1) * n + rem];
    const float ml = omb[18 * om_plane + (long)k * n + rem];
    const float mk = 1.f / (1.f + __expf(-ml));
    const float hf = (float)(ho * p.stride - 1 + k / 3) + dh;
    const float wf = (float)(wo * p.stride - 1 + k % 3) + dw;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (hf > -1.f && wf > -1.f && hf < (float)p.H && wf < (float)p.W) {
      const int h0 = (int)floorf(hf), w0 = (int)floorf(wf);
      const float lh = hf - (float)h0, lw = wf - (float)w0;
      const float wgt[4] = {(1.f - lh) * (1.f - lw), (1.f - lh) * lw, lh * (1.f - lw), lh * lw};
      const half_t* xb = p.x + (long)b * p.H * p.W * p.C + c0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int hh = h0 + (q >> 1), ww = w0 + (q & 1);
        if (hh >= 0 && hh <= p.H - 1 && ww >= 0 && ww <= p.W - 1) {
          half8 v = *(const half8*)(xb + ((long)hh * p.W + ww) * p.C);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += wgt[q] * (float)v[j];
        }
      }
    }
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)(acc[j] * mk);
    *(half8*)(p.cols + (pos * 9 + k) * p.C + c0) = o;
  }
}

extern "C" int mq_dcn_im2col_fwd(const void* x, const float* om, void* cols, int B, int H, int W, int C, int oH, int oW,
                                 int stride, void* stream) {
  if (B <= 0) return 0;
  if (C % 8 || C > 2048 || 256 % (C / 8)) return -1;
  DcnParams p;
  p.x = (const half_t*)x; p.om = om; p.cols = (half_t*)cols;
  p.B = B; p.H = H; p.W = W; p.C = C; p.oH = oH; p.oW = oW; p.stride = stride;
  p.Ho = (H + 2 - 3) / stride + 1; p.Wo = (W + 2 - 3) / stride + 1;
  if ((long)18 * p.Ho * p.Wo > (long)18 * oH * oW) return -2;       // flat reads must stay inside the buffer
  long total = (long)B * p.Ho * p.Wo * 9;
  int ppb = 256 / (C / 8);
  long blocks = (total + ppb - 1) / ppb;
  if (blocks > 256 * 16) blocks = 256 * 16;
  blocks = (blocks + 7) / 8 * 8;                  // whole groups of 8: one block per XCD
  hipLaunchKernelGGL(dcn_im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}
