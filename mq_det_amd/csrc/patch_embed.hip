// mq_patch_embed_fwd: Swin PatchEmbed (4x4 / stride-4 convolution 3 -> C as a 48 -> C projection, swint.py:447-471) + patch_embed.norm
// + the first block's norm1 in ONE pass over the pixels -- gfx950, round 4.
//
//   x32[b, ph, pw, :] = LN_0( W . patch(b, ph, pw) + bias )                fp32 residual stream of Swin stage 1
//   h1 [b, ph, pw, :] = LN_1( x32[b, ph, pw, :] )                          16-bit operand of the first block's attention
//
// Before: permute / reshape copies of the image (3 elementwise launches, 91 us at B = 8), a library GEMM with K = 48 (157 us: 154 MB of
// traffic at 1 TB/s) and two LayerNorm launches (54 + 52 us) = 354 us on the timeline of a step (profiles/r04_call2_timeline_tail.txt
// lists the same run).  Here: the pixels are read once (NHWC: a patch row is 12 contiguous values), the projection is 2 x C/16 MFMAs per
// 16 patches with the weights resident in registers, both LayerNorms happen on the accumulators; HBM traffic = pixels in (51 MB) + x32
// (206 MB) + h1 (103 MB) out.
//
// MFMA layout (out^T = W . patch^T, like the S^T kernels): A = weight rows (channel l & 15 of a 16-channel block, k-slots 8 (l >> 4) ..),
// B = patch^T (patch l & 15 of the block of 16, same k-slots), C: lane l holds channels 4 (l >> 4) + r of patch l & 15 -- the 96 (192)
// channels of one patch live in the 4 lanes l15 + 16 * {0..3}: a LayerNorm statistic is a per-lane sum + 2 shuffles.
// k order: the 48 inputs of a patch are 12 chunks of 4 values (image row py = chunk / 3, values 4 (chunk % 3) .. of its 12); chunk c sits
// in k-step c / 8, k-slots 4 (c % 8) .. -- every fragment half is ONE 8-byte load; the host packs the weight matrix to match
// (mq_det_amd/ops.py patch_embed_pack: [C, 64], k 48..63 zero).
#include "common.h"

MQ_NAMESPACE_BEGIN

struct PatchEmbedParams {
  const void* img;         // F32 = false: [B, Hi, Wi, 3] 16-bit (channels-last image); F32 = true: [B, 3, Hi, Wi] fp32 (the caller's tensor as it is)
  const half_t* w;         // [C, 64] packed projection weight (k order of the image layout, ops.patch_embed_pack)
  const float* bias;       // [C]
  const float* g0; const float* b0;    // patch_embed.norm
  const float* g1; const float* b1;    // layers.0.blocks.0.norm1
  float* x32;              // [B, H * W, C]
  half_t* h1;              // [B, H * W, C]
  int B, Hi, Wi, H, W;     // H = Hi / 4, W = Wi / 4
  long blocks;             // 16-patch blocks: B * H * ceil(W / 16)
  int wblk;                // ceil(W / 16)
  float eps;
};

namespace { constexpr int PE_NW = 4, PE_PER = 4; }         // waves per workgroup, 16-patch blocks per wave

// F32: the pixels are the caller's fp32 NCHW tensor, rounded to the operand type here (what `images.to(dtype)` did in a pass of its own, followed
// by a channels-last copy: two more launches and 150 MB of traffic at B = 8).  k order of that layout: chunk c = (channel c / 4, image row
// c % 4) = 4 consecutive pixels of one plane = ONE 16-byte load; the weight matrix in that order is the conv weight [C, 3, 4, 4] flattened.
template <int C, bool F32>
__global__ __launch_bounds__(64 * PE_NW) void patch_embed_kernel(PatchEmbedParams p) {
  constexpr int CB = C / 16, XP = C + 4, HP = C + 8;        // LDS row pitches (floats / 16-bit values) of the output transpose
  __shared__ __attribute__((aligned(16))) float xs_all[PE_NW * 16 * XP + PE_NW * 16 * HP * (int)sizeof(half_t) / 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  // weights: A fragments of every (channel block, k-step), resident for all blocks of this wave
  half8 wf[CB][2];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int st = 0; st < 2; ++st) wf[cb][st] = *(const half8*)(p.w + (long)(cb * 16 + l15) * 64 + st * 32 + lg * 8);
  float bia[CB][4], ga0[CB][4], be0[CB][4], ga1[CB][4], be1[CB][4];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = cb * 16 + 4 * lg + r;
      bia[cb][r] = p.bias[ch]; ga0[cb][r] = p.g0[ch]; be0[cb][r] = p.b0[ch]; ga1[cb][r] = p.g1[ch]; be1[cb][r] = p.b1[ch];
    }
  const long first = ((long)blockIdx.x * PE_NW + wave) * PE_PER;
  const long pitch = F32 ? (long)p.Wi : (long)p.Wi * 3;    // values per image row (F32: of one plane)
  const long plane = (long)p.Hi * p.Wi;
#pragma unroll 1
  for (int it = 0; it < PE_PER; ++it) {
    const long blk = first + it;
    if (blk >= p.blocks) return;
    const int wb = (int)(blk % p.wblk);
    const long bh = blk / p.wblk;                          // b * H + ph
    const int ph = (int)(bh % p.H), b = (int)(bh / p.H);
    const int pw = min(wb * 16 + l15, p.W - 1);            // patches beyond the row: computed on the last one, not stored
    // B fragments: k-step st, slots 4 (2 lg + half) .. = chunk c = 8 st + 2 lg + half (c < 12)
    half8 bf[2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int c = 8 * st + 2 * lg + hf;
        half4 v = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        if constexpr (F32) {
          if (c < 12) {
            const float4_ f = *(const float4_*)((const float*)p.img + ((long)b * 3 + (c >> 2)) * plane + (long)(4 * ph + (c & 3)) * pitch + (long)pw * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (half_t)f[e];
          }
        } else {
          const half_t* base = (const half_t*)p.img + ((long)b * p.Hi + 4 * ph) * pitch + (long)pw * 12;
          if (c < 12) v = *(const half4*)(base + (long)(c / 3) * pitch + (c % 3) * 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) bf[st][4 * hf + e] = v[e];
      }
    float4_ acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      acc[cb] = (float4_){0.f, 0.f, 0.f, 0.f};
      acc[cb] = mfma16(wf[cb][0], bf[0], acc[cb]);
      acc[cb] = mfma16(wf[cb][1], bf[1], acc[cb]);
    }
    // ---- + bias, LayerNorm 0 (two-pass statistics over the C channels of this lane's patch: 4 lanes x CB x 4 values)
    float sum = 0.f;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[cb][r] += bia[cb][r]; sum += acc[cb][r]; }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.f / C);
    float var = 0.f;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc[cb][r] - mean; var += d * d; }
    var += __shfl_xor(var, 16);
    var += __shfl_xor(var, 32);
    const float rstd = rsqrtf(var * (1.f / C) + p.eps);
    float sum1 = 0.f;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[cb][r] = (acc[cb][r] - mean) * rstd * ga0[cb][r] + be0[cb][r]; sum1 += acc[cb][r]; }
    // ---- LayerNorm 1 on the fp32 stream values
    sum1 += __shfl_xor(sum1, 16);
    sum1 += __shfl_xor(sum1, 32);
    const float mean1 = sum1 * (1.f / C);
    float var1 = 0.f;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc[cb][r] - mean1; var1 += d * d; }
    var1 += __shfl_xor(var1, 16);
    var1 += __shfl_xor(var1, 32);
    const float rstd1 = rsqrtf(var1 * (1.f / C) + p.eps);
    // ---- the 16 patches of a block are 16 consecutive rows of x32 / h1: transpose through LDS (this wave's own slice, no barrier) and
    // write whole rows, 16 bytes per lane -- 1 KB per store instruction instead of 64-byte (x32) / 32-byte (h1) pieces
    float* xs = xs_all + wave * (16 * XP);
    half_t* hs = (half_t*)(xs_all + PE_NW * 16 * XP) + wave * (16 * HP);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      *(float4_*)(xs + l15 * XP + cb * 16 + 4 * lg) = acc[cb];
      half4 hv;
#pragma unroll
      for (int r = 0; r < 4; ++r) hv[r] = (half_t)((acc[cb][r] - mean1) * rstd1 * ga1[cb][r] + be1[cb][r]);
      *(half4*)(hs + l15 * HP + cb * 16 + 4 * lg) = hv;
    }
    wave_lds_fence();
    const int nvalid = min(16, p.W - wb * 16);             // patches of this block inside the row
    const long tok0 = bh * p.W + wb * 16;
    float* xo = p.x32 + tok0 * C;
    half_t* ho = p.h1 + tok0 * C;
#pragma unroll
    for (int i = 0; i < 16 * C / 4 / 64; ++i) {
      const int f = lane + i * 64, row = f / (C / 4), c4 = f - row * (C / 4);
      if (row < nvalid) *(float4_*)(xo + (long)row * C + c4 * 4) = *(const float4_*)(xs + row * XP + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < 16 * C / 8 / 64; ++i) {
      const int f = lane + i * 64, row = f / (C / 8), c8 = f - row * (C / 8);
      if (row < nvalid) *(half8*)(ho + (long)row * C + c8 * 8) = *(const half8*)(hs + row * HP + c8 * 8);
    }
    wave_lds_fence();                                      // the slice is rewritten by the next block of this wave
  }
}

// img: img_f32 == 0: [B, Hi, Wi, 3] 16-bit channels-last pixels; img_f32 != 0: [B, 3, Hi, Wi] fp32 (rounded to the operand type here).  Hi, Wi
// multiples of 4.  w [C, 64] packed for that layout (ops.patch_embed_pack), bias / g0 / b0 / g1 / b1 fp32 [C]; x32 [B, (Hi/4) * (Wi/4), C]
// fp32 and h1 (same shape, 16-bit) out.  C = 96 (Swin-T) or 192 (Swin-L).  -1 otherwise.
extern "C" int MQ_SYM(mq_patch_embed_fwd)(const void* img, int img_f32, const void* w, const float* bias, const float* g0, const float* b0,
                                          const float* g1, const float* b1, float* x32, void* h1, int B, int Hi, int Wi, int C, float eps,
                                          void* stream) {
  if (B <= 0 || Hi <= 0 || Wi <= 0) return 0;
  if ((Hi & 3) || (Wi & 3) || (C != 96 && C != 192)) return -1;
  PatchEmbedParams p;
  p.img = img; p.w = (const half_t*)w; p.bias = bias; p.g0 = g0; p.b0 = b0; p.g1 = g1; p.b1 = b1;
  p.x32 = x32; p.h1 = (half_t*)h1; p.B = B; p.Hi = Hi; p.Wi = Wi; p.H = Hi / 4; p.W = Wi / 4; p.eps = eps;
  p.wblk = (p.W + 15) / 16;
  p.blocks = (long)B * p.H * p.wblk;
  const long per_wg = (long)PE_NW * PE_PER;
  const dim3 grid((unsigned)((p.blocks + per_wg - 1) / per_wg)), block(64 * PE_NW);
  hipStream_t s = (hipStream_t)stream;
  if (C == 96 && img_f32) hipLaunchKernelGGL((patch_embed_kernel<96, true>), grid, block, 0, s, p);
  else if (C == 96) hipLaunchKernelGGL((patch_embed_kernel<96, false>), grid, block, 0, s, p);
  else if (img_f32) hipLaunchKernelGGL((patch_embed_kernel<192, true>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((patch_embed_kernel<192, false>), grid, block, 0, s, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

MQ_NAMESPACE_END
