// mq_roi_align_fwd: ROIAlign (legacy and aligned) for the vision-query extraction path, gfx950.
//
// Reference: maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:16-123 (bilinear_interpolate, RoIAlignForward: the legacy
// operator, `layers/roi_align.py` ROIAlign) and torchvision.ops.roi_align(aligned=True) (`ROIAlignV2`,
// layers/roi_align.py:71-81 -- what `Pooler(use_v2=True)` of generalized_vl_rcnn_new.py:108-121 uses): aligned shifts the
// scaled box by -0.5 and drops the "malformed ROIs are 1x1" clamp.
//
// MI355X shape of the problem: a few hundred boxes x 256 channels x 7 x 7 bins -- latency / HBM-gather bound, no
// contraction.  The reference indexes NCHW (one thread per output element, channel stride H*W: 4 uncoalesced corner loads
// per sample); here the feature map is read through arbitrary element strides, and for the product's NHWC fp16 pyramid
// the 64 lanes of a wave are 64 consecutive CHANNELS of one (roi, bin): every corner load is one coalesced 128-byte line,
// the sampling geometry (identical for the whole wave) is computed once per wave in scalar registers.
//   feat   : element (n, c, y, x) at feat + n*sn + c*sc + y*sh + x*sw   (fp16, or fp32 when feat_f32)
//   rois   : [R, 5] fp32 (batch index, x1, y1, x2, y2) in image coordinates
//   out    : [R, C, PH, PW] fp32, or [R, C] fp32 = mean over the bins when reduce_mean (what extract_query keeps,
//            generalized_vl_rcnn_new.py:263)
#include "common.h"

MQ_NAMESPACE_BEGIN

template <typename TF>
__global__ __launch_bounds__(256) void roi_align_kernel(const TF* __restrict__ feat, const float* __restrict__ rois,
                                                        float* __restrict__ out, int R, int C, int H, int W, long sn, long sc,
                                                        long sh, long sw, int PH, int PW, float scale, int sampling, int aligned,
                                                        int reduce_mean) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cblocks = (C + 63) >> 6;
  const long task = (long)blockIdx.x * 4 + wave;               // (roi, bin, channel block); bins folded when reduce_mean
  const int nbin = reduce_mean ? 1 : PH * PW;
  if (task >= (long)R * nbin * cblocks) return;
  const int cb = task % cblocks;
  const int bin = (task / cblocks) % nbin;
  const int r = task / ((long)cblocks * nbin);
  const int c = cb * 64 + lane;
  const float* roi = rois + (long)r * 5;
  const int n = (int)roi[0];
  const float off = aligned ? 0.5f : 0.f;
  const float x1 = roi[1] * scale - off, y1 = roi[2] * scale - off, x2 = roi[3] * scale - off, y2 = roi[4] * scale - off;
  float rw = x2 - x1, rh = y2 - y1;
  if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
  const float bh = rh / (float)PH, bw = rw / (float)PW;
  const int gh = sampling > 0 ? sampling : (int)ceilf(rh / (float)PH);
  const int gw = sampling > 0 ? sampling : (int)ceilf(rw / (float)PW);
  const float count = aligned ? fmaxf((float)(gh * gw), 1.f) : (float)(gh * gw);
  const TF* fb = feat + (long)n * sn + (long)c * sc;
  const bool live = c < C;
  float total = 0.f;
  const int b0 = reduce_mean ? 0 : bin, b1 = reduce_mean ? PH * PW : bin + 1;
  for (int b = b0; b < b1; ++b) {
    const int ph = b / PW, pw = b % PW;
    float acc = 0.f;
    for (int iy = 0; iy < gh; ++iy) {
      float y = y1 + ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        float x = x1 + pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
        float yy = y;
        if (yy < -1.0f || yy > (float)H || x < -1.0f || x > (float)W) continue;       // wave-uniform
        if (yy <= 0.f) yy = 0.f;
        if (x <= 0.f) x = 0.f;
        int yl = (int)yy, xl = (int)x, yh, xh;
        if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
        if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
        const float ly = yy - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
        if (live) {
          const float v1 = (float)fb[yl * sh + xl * sw], v2 = (float)fb[yl * sh + xh * sw];
          const float v3 = (float)fb[yh * sh + xl * sw], v4 = (float)fb[yh * sh + xh * sw];
          acc += hy * hx * v1 + hy * lx * v2 + ly * hx * v3 + ly * lx * v4;
        }
      }
    }
    acc = (gh > 0 && gw > 0) ? acc / count : 0.f;
    if (!reduce_mean) {
      if (live) out[(((long)r * C + c) * PH + ph) * PW + pw] = acc;
    } else {
      total += acc;
    }
  }
  if (reduce_mean && live) out[(long)r * C + c] = total / (float)(PH * PW);
}

extern "C" int MQ_SYM(mq_roi_align_fwd)(const void* feat, int feat_f32, const float* rois, float* out, int R, int C, int H, int W,
                                long sn, long sc, long sh, long sw, int PH, int PW, float spatial_scale, int sampling_ratio,
                                int aligned, int reduce_mean, void* stream) {
  if (R <= 0 || C <= 0) return 0;
  if (PH <= 0 || PW <= 0) return -1;
  const long tasks = (long)R * (reduce_mean ? 1 : PH * PW) * ((C + 63) / 64);
  const dim3 grid((unsigned)((tasks + 3) / 4));
  if (feat_f32)
    hipLaunchKernelGGL(roi_align_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)feat, rois, out, R, C, H, W,
                       sn, sc, sh, sw, PH, PW, spatial_scale, sampling_ratio, aligned, reduce_mean);
  else
    hipLaunchKernelGGL(roi_align_kernel<half_t>, grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)feat, rois, out, R, C, H,
                       W, sn, sc, sh, sw, PH, PW, spatial_scale, sampling_ratio, aligned, reduce_mean);
  MQ_CHECK_LAUNCH();
  return 0;
}

MQ_NAMESPACE_END
