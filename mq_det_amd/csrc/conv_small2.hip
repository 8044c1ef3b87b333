// mq_conv3x3_nchw32_v2_fwd: the offset / mask conv of conv_small.hip (same operator, arguments, tiling, MFMA order -- bit-identical
// results) with a different LOAD schedule.  In conv_small.hip the window fill guards every load with the image-border test
// (`v = 0; if (inside) v = load`): hipcc emits a branch per load and a `s_waitcnt vmcnt(0)` after every second one (ISA: 4 of the 8
// loads of a batch are waited on at once) -- ~16 exposed memory round trips per tile against ~2 us of MFMA work, on a kernel that is
// 1.07 ms of the MQ-GLIP step.  Here every address is clamped into the image and loaded unconditionally, all
// (<= 12) chunks of a thread are in flight before the first LDS store, the border test only selects zero afterwards; the weight
// prefetch is unconditional as well.  Same lesson as the MSDeformAttn gather (2.38 -> 1.03 ms) and layernorm2.hip.
// Default since round 3 (KERNELS["OFFSET_CONV_VARIANT"] = 2): +4.4 % end to end on its own (profiles/r03_call1_switch_ab.txt), equal
// outputs to mq_conv3x3_nchw32_fwd on the device and through tests/simt.
#include "common.h"

// SPLIT-PRECISE build (-DMQ_F32, round 6): the window and the weight slices are PLANAR in LDS -- hi = fp16(x) and lo = fp16((x - hi) 2^11) planes
// with the fp16 build's pitch, the same bytes as the fp32 tiles -- split ONCE by the thread that stages a chunk; the k-loop reads one 16-byte
// fragment per plane and issues three fp16 MFMAs (mfma16_split).  (Splitting inside mfma16 cost ~100 VALU instructions per 12 MFMAs here: every
// fragment feeds only two.)
#if defined(MQ_F32) && !defined(MQ_F32_EXACT)
#define MQ_CS_SPLIT 1
#else
#define MQ_CS_SPLIT 0
#endif

MQ_NAMESPACE_BEGIN

struct ConvSmall2Params {
  const half_t* x; const half_t* w; const half_t* bias; float* out;
  long x_bs;
  int B, H, W, C, N, tiles_x, tiles_y, tiles_total;
};

namespace {
constexpr int CS2_PH = 8, CS2_PW = 16, CS2_WH = CS2_PH + 2, CS2_WW = CS2_PW + 2;
}

__global__ __launch_bounds__(256) void conv3x3_small2_kernel(ConvSmall2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int C = p.C;
  const int CP = C > 128 ? C / 2 : C;                        // channels per pass (C % 64 == 0 when C > 128)
  const int XP = CP + 16;                                    // window / weight row pitch (halfs): 8 rows span all 64 banks
  half_t* Win = (half_t*)smem;                               // [CS2_WH * CS2_WW][XP]
  half_t* Ws = Win + CS2_WH * CS2_WW * XP;                     // [2][32][XP]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  // XCD-aware tile order (see conv_igemm.hip): XCD x owns a contiguous range of patches -> halos re-used from its L2
  const int tpx = (p.tiles_total + 7) >> 3;
  const int tile = (blockIdx.x & 7) * tpx + (blockIdx.x >> 3);
  if (tile >= p.tiles_total) return;
  const int b = tile / (p.tiles_x * p.tiles_y), trem = tile % (p.tiles_x * p.tiles_y);
  const int ho0 = (trem / p.tiles_x) * CS2_PH, wo0 = (trem % p.tiles_x) * CS2_PW;
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
    auto w_issue = [&](int tap) {                           // unconditional: chunks beyond the slice re-read its last chunk
#pragma unroll
      for (int i = 0; i < WMAX; ++i) {
        const int c = min(tid + i * 256, 32 * cpr - 1);
        wreg[i] = *(const half8*)(p.w + (long)(c / cpr) * K + tap * C + c0 + (c % cpr) * 8);
      }
    };
    auto w_commit = [&](int buf) {
#pragma unroll
      for (int i = 0; i < WMAX; ++i) {
        const int c = tid + i * 256;
#if MQ_CS_SPLIT
        if (c < 32 * cpr) {
          const mq_split8 sp = mq_split(wreg[i]);
          _Float16* d = (_Float16*)Ws + (buf * 32 + c / cpr) * XP + (c % cpr) * 8;
          *(mq_h16x8*)d = sp.hi;
          *(mq_h16x8*)(d + 2 * 32 * XP) = sp.lo;
        }
#else
        if (c < 32 * cpr) *(half8*)(Ws + (buf * 32 + c / cpr) * XP + (c % cpr) * 8) = wreg[i];
#endif
      }
    };
    w_issue(0);
    // ---- input window of this pass -> LDS: every chunk of this thread (<= 12) is loaded from an address clamped into the image,
    // all loads in flight before the first store; pixels outside the image are zeroed by a select on the way to LDS
    {
      constexpr int NU = (CS2_WH * CS2_WW * 16 + 255) / 256;   // 12 chunks per thread at 128 channels per pass
      const int total = CS2_WH * CS2_WW * cpr;
      half8 v[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int c = min(u * 256 + tid, total - 1);
        const int px = c / cpr, ch = c - px * cpr;
        const int hh = min(max(ho0 - 1 + px / CS2_WW, 0), p.H - 1), ww = min(max(wo0 - 1 + px % CS2_WW, 0), p.W - 1);
        v[u] = *(const half8*)(xb + ((long)hh * p.W + ww) * C + c0 + ch * 8);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int c = u * 256 + tid;
        const int cc = min(c, total - 1);
        const int px = cc / cpr, ch = cc - px * cpr;
        const int hh = ho0 - 1 + px / CS2_WW, ww = wo0 - 1 + px % CS2_WW;
        const bool inside = hh >= 0 && hh < p.H && ww >= 0 && ww < p.W;
#if MQ_CS_SPLIT
        if (c < total) {
          const mq_split8 sp = mq_split(inside ? v[u] : zero8());
          _Float16* d = (_Float16*)Win + px * XP + ch * 8;
          *(mq_h16x8*)d = sp.hi;
          *(mq_h16x8*)(d + CS2_WH * CS2_WW * XP) = sp.lo;
        }
#else
        if (c < total) *(half8*)(Win + px * XP + ch * 8) = inside ? v[u] : zero8();
#endif
      }
    }
    w_commit(0);
    __syncthreads();
    // this wave: patch rows 2*wave, 2*wave + 1 (block i = patch row, l15 = column inside the patch)
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) w_issue(tap + 1);
      const int dy = tap / 3, dx = tap - dy * 3;
#if MQ_CS_SPLIT
      const _Float16* a0 = (const _Float16*)Win + ((2 * wave + dy) * CS2_WW + l15 + dx) * XP + lg * 8;
      const _Float16* b0 = (const _Float16*)Ws + ((tap & 1) * 32 + l15) * XP + lg * 8;
      for (int kk = 0; kk < CP / 32; ++kk) {
        mq_split8 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[i].hi = *(const mq_h16x8*)(a0 + i * CS2_WW * XP + kk * 32);
          af[i].lo = *(const mq_h16x8*)(a0 + CS2_WH * CS2_WW * XP + i * CS2_WW * XP + kk * 32);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf[j].hi = *(const mq_h16x8*)(b0 + j * 16 * XP + kk * 32);
          bf[j].lo = *(const mq_h16x8*)(b0 + 2 * 32 * XP + j * 16 * XP + kk * 32);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma16_split(af[i], bf[j], acc[i][j]);
      }
#else
      const half_t* a0 = Win + ((2 * wave + dy) * CS2_WW + l15 + dx) * XP + lg * 8;
      const half_t* b0 = Ws + ((tap & 1) * 32 + l15) * XP + lg * 8;
      for (int kk = 0; kk < CP / 32; ++kk) {
        half8 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *(const half8*)(a0 + i * CS2_WW * XP + kk * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *(const half8*)(b0 + j * 16 * XP + kk * 32);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
      }
#endif
      if (tap + 1 < 9) w_commit((tap + 1) & 1);
      __syncthreads();                                       // also: window + weights free for the next pass
    }
  }

  // ---- epilogue: + bias, fp32 NCHW.  C layout: row = position 4*lg + r of patch row i, col = channel j*16 + l15
  float* Os = (float*)smem;                                  // [32 ch][CS2_PH * CS2_PW + 4] (window is dead: last barrier passed)
  constexpr int OP = CS2_PH * CS2_PW + 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = j * 16 + l15;
    const float bv = (p.bias && n < p.N) ? (float)p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) Os[n * OP + (2 * wave + i) * CS2_PW + 4 * lg + r] = acc[i][j][r] + bv;
  }
  __syncthreads();
  float* ob = p.out + (long)b * p.N * p.H * p.W;
  for (int c = tid; c < p.N * CS2_PH * CS2_PW; c += 256) {
    const int n = c / (CS2_PH * CS2_PW), pos = c % (CS2_PH * CS2_PW);
    const int ho = ho0 + pos / CS2_PW, wo = wo0 + pos % CS2_PW;
    if (ho < p.H && wo < p.W) ob[((long)n * p.H + ho) * p.W + wo] = Os[n * OP + pos];
  }
}

// x [B,H,W,C] fp16 NHWC (batch stride x_bs, C % 32 == 0, C <= 256), w [32, 9*C] fp16 (k = tap*C + c, rows >= N zero),
// bias [N] fp16 or NULL -> out [B, N, H, W] fp32 (NCHW), stride 1, pad 1.
extern "C" int MQ_SYM(mq_conv3x3_nchw32_v2_fwd)(const void* x, const void* w, const void* bias, float* out, int B, int H, int W, int C,
                                     long x_bs, int N, void* stream) {
  if (B <= 0) return 0;
  if (C % 32 || C > 256 || (C > 128 && C % 64) || N < 1 || N > 32) return -1;
  ConvSmall2Params p;
  p.x = (const half_t*)x; p.w = (const half_t*)w; p.bias = (const half_t*)bias; p.out = out;
  p.x_bs = x_bs; p.B = B; p.H = H; p.W = W; p.C = C; p.N = N;
  p.tiles_y = (H + CS2_PH - 1) / CS2_PH; p.tiles_x = (W + CS2_PW - 1) / CS2_PW;
  p.tiles_total = B * p.tiles_y * p.tiles_x;
  const int CP = C > 128 ? C / 2 : C;
  const size_t tiles = (size_t)(CS2_WH * CS2_WW + 2 * 32) * (CP + 16) * sizeof(half_t);
  const size_t ostage = (size_t)32 * (CS2_PH * CS2_PW + 4) * sizeof(float);
  const size_t smem = tiles > ostage ? tiles : ostage;
  static MqMaxPerDevice attr_set;
  if (attr_set.need(smem)) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_small2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set.done(smem);
  }
  hipLaunchKernelGGL(conv3x3_small2_kernel, dim3((unsigned)(8 * ((p.tiles_total + 7) / 8))), dim3(256), smem, (hipStream_t)stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

// (Round 3 tried a third version without the weight tile: B fragments straight from global memory through a six-step register ring,
// the window the only LDS tile, one barrier pair per channel pass, three workgroups per CU.  Equal results, but 0.123 ms against this
// kernel's 0.055 ms on the P3 level at B = 8 (profiles/r03_call7_microbench_dyconv.json): 72 dependent 16-byte loads per wave and pass
// from L2 are slower than ten barriers.  Removed.)

MQ_NAMESPACE_END
