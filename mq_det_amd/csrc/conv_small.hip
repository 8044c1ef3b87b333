// mq_conv3x3_nchw32_fwd: 3x3 convolution (pad 1, stride 1) with FEW output channels (N <= 32), NHWC fp16 in,
// fp32 NCHW out -- the 27-channel offset / mask conv of every DyConv level (reference rpn/vldyhead.py:186,214:
// nn.Conv2d(256, 27, 3) followed by the fp32 offset / mask split), 30 launches per forward.
//
// The generic implicit-GEMM kernel (conv_igemm.hip) re-gathers the 9 taps of every position from global memory
// (590 KB through the vector-memory path per 128 positions) and spends ~1500 cycles per 32-wide k-step on a 32-column
// tile that keeps the matrix pipe idle: 48 us per tile.  Here a workgroup owns an 8 x 16 patch of output positions:
//   * the (8+2) x (16+2) input window (zero-padded at the image border) is loaded ONCE into LDS with coalesced 16-byte
//     loads -- every input byte crosses the vector-memory path once, the 9 taps read it from LDS.  Channels are walked in
//     passes of <= 128 (window 52 KB + weights 2 x 9 KB): two workgroups fit a CU, so one computes while the other
//     waits for its window (a single 95 KB window for all 256 channels left the CU idle during every load);
//   * the weight slice of one (pass, tap) ([32, 128], 8 KB) is double-buffered in LDS, prefetched through registers
//     while the previous tap is on the MFMAs; one barrier per tap (18 per tile instead of 72);
//   * 4 waves x (32 positions x 32 channels): A fragments are ds_read_b128 straight out of the window (a 16-row MFMA
//     block is one patch row, pitch C+16 halfs: conflict-free), v_mfma_f32_16x16x32_f16, fp32 accumulation over
//     K = 9*C in a fixed order;
//   * the epilogue adds the bias and writes fp32 NCHW directly (what the DCN kernels read), replacing a separate
//     permute + float() pass.
#include "common.h"

MQ_NAMESPACE_BEGIN

struct ConvSmallParams {
  const half_t* x; const half_t* w; const half_t* bias; float* out;
  long x_bs;
  int B, H, W, C, N, tiles_x, tiles_y, tiles_total;
};

namespace {
constexpr int CS_PH = 8, CS_PW = 16, CS_WH = CS_PH + 2, CS_WW = CS_PW + 2;
}

__global__ __launch_bounds__(256) void conv3x3_small_kernel(ConvSmallParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int C = p.C;
  const int CP = C > 128 ? C / 2 : C;                        // channels per pass (C % 64 == 0 when C > 128)
  const int XP = CP + 16;                                    // window / weight row pitch (halfs): 8 rows span all 64 banks
  half_t* Win = (half_t*)smem;                               // [CS_WH * CS_WW][XP]
  half_t* Ws = Win + CS_WH * CS_WW * XP;                     // [2][32][XP]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  // XCD-aware tile order (see conv_igemm.hip): XCD x owns a contiguous range of patches -> halos re-used from its L2
  const int tpx = (p.tiles_total + 7) >> 3;
  const int tile = (blockIdx.x & 7) * tpx + (blockIdx.x >> 3);
  if (tile >= p.tiles_total) return;
  const int b = tile / (p.tiles_x * p.tiles_y), trem = tile % (p.tiles_x * p.tiles_y);
  const int ho0 = (trem / p.tiles_x) * CS_PH, wo0 = (trem % p.tiles_x) * CS_PW;
  const int cpr = CP / 8;                                    // 16-byte chunks per pixel / weight row and pass
  const half_t* xb = p.x + (long)b * p.x_bs;
  const int K = 9 * C;
  constexpr int WMAX = 2;                                    // weight chunks per thread (32 rows x <= 128 channels)
  half8 wreg[WMAX];

  float4_ acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (float4_){0.f, 0.f, 0.f, 0.f};

  for (int c0 = 0; c0 < C; c0 += CP) {                       // channel pass
    auto w_issue = [&](int tap) {
#pragma unroll
      for (int i = 0; i < WMAX; ++i) {
        const int c = tid + i * 256;
        if (c < 32 * cpr) wreg[i] = *(const half8*)(p.w + (long)(c / cpr) * K + tap * C + c0 + (c % cpr) * 8);
      }
    };
    auto w_commit = [&](int buf) {
#pragma unroll
      for (int i = 0; i < WMAX; ++i) {
        const int c = tid + i * 256;
        if (c < 32 * cpr) *(half8*)(Ws + (buf * 32 + c / cpr) * XP + (c % cpr) * 8) = wreg[i];
      }
    };
    w_issue(0);
    // ---- input window of this pass -> LDS (zero outside the image); 8 loads in flight per thread, then 8 LDS stores
    for (int base = 0; base < CS_WH * CS_WW * cpr; base += 8 * 256) {
      half8 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = base + u * 256 + tid;
        const int px = c / cpr, ch = c - px * cpr;
        const int hh = ho0 - 1 + px / CS_WW, ww = wo0 - 1 + px % CS_WW;
        v[u] = zero8();
        if (c < CS_WH * CS_WW * cpr && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
          v[u] = *(const half8*)(xb + ((long)hh * p.W + ww) * C + c0 + ch * 8);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = base + u * 256 + tid;
        const int px = c / cpr, ch = c - px * cpr;
        if (c < CS_WH * CS_WW * cpr) *(half8*)(Win + px * XP + ch * 8) = v[u];
      }
    }
    w_commit(0);
    __syncthreads();
    // this wave: patch rows 2*wave, 2*wave + 1 (block i = patch row, l15 = column inside the patch)
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) w_issue(tap + 1);
      const int dy = tap / 3, dx = tap - dy * 3;
      const half_t* a0 = Win + ((2 * wave + dy) * CS_WW + l15 + dx) * XP + lg * 8;
      const half_t* b0 = Ws + ((tap & 1) * 32 + l15) * XP + lg * 8;
      for (int kk = 0; kk < CP / 32; ++kk) {
        half8 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *(const half8*)(a0 + i * CS_WW * XP + kk * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *(const half8*)(b0 + j * 16 * XP + kk * 32);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
      }
      if (tap + 1 < 9) w_commit((tap + 1) & 1);
      __syncthreads();                                       // also: window + weights free for the next pass
    }
  }

  // ---- epilogue: + bias, fp32 NCHW.  C layout: row = position 4*lg + r of patch row i, col = channel j*16 + l15
  float* Os = (float*)smem;                                  // [32 ch][CS_PH * CS_PW + 4] (window is dead: last barrier passed)
  constexpr int OP = CS_PH * CS_PW + 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = j * 16 + l15;
    const float bv = (p.bias && n < p.N) ? (float)p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) Os[n * OP + (2 * wave + i) * CS_PW + 4 * lg + r] = acc[i][j][r] + bv;
  }
  __syncthreads();
  float* ob = p.out + (long)b * p.N * p.H * p.W;
  for (int c = tid; c < p.N * CS_PH * CS_PW; c += 256) {
    const int n = c / (CS_PH * CS_PW), pos = c % (CS_PH * CS_PW);
    const int ho = ho0 + pos / CS_PW, wo = wo0 + pos % CS_PW;
    if (ho < p.H && wo < p.W) ob[((long)n * p.H + ho) * p.W + wo] = Os[n * OP + pos];
  }
}

// x [B,H,W,C] fp16 NHWC (batch stride x_bs, C % 32 == 0, C <= 256), w [32, 9*C] fp16 (k = tap*C + c, rows >= N zero),
// bias [N] fp16 or NULL -> out [B, N, H, W] fp32 (NCHW), stride 1, pad 1.
extern "C" int MQ_SYM(mq_conv3x3_nchw32_fwd)(const void* x, const void* w, const void* bias, float* out, int B, int H, int W, int C,
                                     long x_bs, int N, void* stream) {
  if (B <= 0) return 0;
  if (C % 32 || C > 256 || (C > 128 && C % 64) || N < 1 || N > 32) return -1;
  ConvSmallParams p;
  p.x = (const half_t*)x; p.w = (const half_t*)w; p.bias = (const half_t*)bias; p.out = out;
  p.x_bs = x_bs; p.B = B; p.H = H; p.W = W; p.C = C; p.N = N;
  p.tiles_y = (H + CS_PH - 1) / CS_PH; p.tiles_x = (W + CS_PW - 1) / CS_PW;
  p.tiles_total = B * p.tiles_y * p.tiles_x;
  const int CP = C > 128 ? C / 2 : C;
  const size_t tiles = (size_t)(CS_WH * CS_WW + 2 * 32) * (CP + 16) * sizeof(half_t);
  const size_t ostage = (size_t)32 * (CS_PH * CS_PW + 4) * sizeof(float);
  const size_t smem = tiles > ostage ? tiles : ostage;
  static MqMaxPerDevice attr_set;
  if (attr_set.need(smem)) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set.done(smem);
  }
  hipLaunchKernelGGL(conv3x3_small_kernel, dim3((unsigned)(8 * ((p.tiles_total + 7) / 8))), dim3(256), smem, (hipStream_t)stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

MQ_NAMESPACE_END
