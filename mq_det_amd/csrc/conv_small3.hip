// mq_conv3x3_nchw32_group_fwd: the offset / mask conv of a DyConv layer (vldyhead.py:205-215: ONE 3x3 conv, 256 -> 27 channels, applied to
// every pyramid level) for ALL levels in one launch of persistent workgroups that keep the weights in registers.
//
// conv_small2.hip (one launch per level, five streams) is 0.83 ms of the MQ-GLIP-T step at 10 % of the HBM roofline: per 8 x 16 tile and
// 128-channel pass it runs nine taps with the weight slice of each tap going global -> registers -> LDS -> barrier -- 20 workgroup barriers
// per tile around ~100 cycles of MFMA work each, 147 KB of weights re-read from L2 per tile, and the P5 .. P7 launches (96 / 32 / 8 tiles)
// cannot fill 256 CUs.  Here
//   * one workgroup of 8 waves per CU walks a contiguous run of tiles (all levels form one work list, XCD-contiguous chunks, like
//     mq_dcnv2_group_fwd);
//   * the contraction is split over the waves by CHANNEL: wave w owns channels [32 w, 32 w + 32) and ALL 128 positions of the tile.  Its B
//     fragments -- 9 taps x 2 column blocks x 16 B per lane = 72 VGPRs -- are loaded ONCE per workgroup and stay in registers: no weight
//     tile in LDS, no barrier inside a tile's nine taps, weights read 256 times per launch instead of once per tile and pass;
//   * the window of tile k + 1 (all 256 channels: 12 x 16 B per thread) is in flight in registers while tile k computes -- 92 KB per CU
//     outstanding, what 8 TB/s x ~2 us of latency needs;
//   * the eight partial sums of a position meet in LDS (fp32, fixed order) on the way to the NCHW store.
// Summation order differs from conv_small(2).hip (there: one accumulator over taps and channels), so results agree to fp32 rounding, not
// bit for bit.  KERNELS["OFFSET_CONV_VARIANT"] = 3.
#include "common.h"

MQ_NAMESPACE_BEGIN

namespace {
constexpr int CS3_PH = 8, CS3_PW = 16, CS3_WH = CS3_PH + 2, CS3_WW = CS3_PW + 2;
constexpr int CS3_C = 256, CS3_XP = CS3_C + 16;              // channels; window row pitch (halfs): 8 rows span all 64 banks
constexpr int CS3_OP = CS3_PH * CS3_PW + 4;                  // exchange row pitch (floats)
constexpr int CS3_NW = CS3_C / 32, CS3_NT = 64 * CS3_NW;     // waves, threads per workgroup
constexpr int CS3_MAX_LEVELS = 8;
constexpr int CS3_CPR = CS3_C / 8, CS3_TOTAL = CS3_WH * CS3_WW * CS3_CPR;   // 16-byte chunks per pixel / per window
constexpr int CS3_NU = (CS3_TOTAL + CS3_NT - 1) / CS3_NT;    // window chunks per thread (12)
}

struct ConvLevel {
  const half_t* x; float* out; long x_bs;
  int H, W, tiles_x, tiles_y, first_tile, pad_;
};
struct ConvGroupParams {
  ConvLevel lv[CS3_MAX_LEVELS];
  const half_t* w; const half_t* bias;
  int nl, B, N, tiles_total, tiles_per_xcd, tiles_per_block;
};

struct ConvTile {                                            // uniform (scalar) description of one tile
  const half_t* xb; float* ob;
  int H, W, ho0, wo0;
};

__device__ __forceinline__ ConvTile conv3_tile(const ConvGroupParams& p, int tile) {
  int L = 0;
  while (L + 1 < p.nl && tile >= p.lv[L + 1].first_tile) ++L;
  const ConvLevel& v = p.lv[L];
  const int t = tile - v.first_tile, per = v.tiles_x * v.tiles_y;
  const int b = t / per, trem = t - b * per;
  ConvTile r;
  r.H = v.H; r.W = v.W;
  r.ho0 = (trem / v.tiles_x) * CS3_PH; r.wo0 = (trem % v.tiles_x) * CS3_PW;
  r.xb = v.x + (long)b * v.x_bs;
  r.ob = v.out + (long)b * p.N * v.H * v.W;
  return r;
}

__global__ __launch_bounds__(CS3_NT) void conv3x3_group_kernel(ConvGroupParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* Win = (half_t*)smem;                               // [CS3_WH * CS3_WW][CS3_XP]
  float* Os = (float*)smem;                                  // [8 waves][32 ch][CS3_OP] between the taps and the store
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  // XCD x (= blockIdx & 7) owns tiles [x * tiles_per_xcd, ...): halos re-used from its L2; its workgroups take contiguous runs
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int xend = min(p.tiles_total, (xcd + 1) * p.tiles_per_xcd);
  int tile = xcd * p.tiles_per_xcd + slot * p.tiles_per_block;
  const int tend = min(xend, tile + p.tiles_per_block);
  if (tile >= tend) return;
  constexpr int K = 9 * CS3_C;

  // B fragments of this wave's channel slice: row (output channel) j*16 + l15, k = tap*C + 32*wave + 8*lg .. +7
  half8 wr[9][2];
  {
    const unsigned wlane = (unsigned)(l15 * K + wave * 32 + lg * 8);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int j = 0; j < 2; ++j) wr[tap][j] = *(const half8*)(p.w + (j * 16 * K + tap * CS3_C) + wlane);
  }
  // window chunk u of this thread: pixel px0 + 16 u of the 10 x 18 window, channels 8 ch .. 8 ch + 7 (512 threads = 16 pixels x 32 chunks).
  // MQ_PIN: the per-chunk rows / columns / clamps are a dozen VALU ops each -- recomputed per tile instead of living in ~50 VGPRs next to
  // 72 of weights, 64 of accumulators and 48 of window in flight (hoisted, they spilled 123 registers)
#define MQ_PIN(x) asm volatile("" : "+v"(x))
  static_assert(CS3_NT == 16 * CS3_CPR, "512 threads = 16 pixels x 32 chunks");
  const int ch8 = (tid & (CS3_CPR - 1)) * 8;
  half8 win[CS3_NU];
  auto win_issue = [&](const ConvTile& t) {                 // every address clamped into the image, all loads in flight
    int px0 = tid >> 5;
    MQ_PIN(px0);
    px0 &= 15;
#pragma unroll
    for (int u = 0; u < CS3_NU; ++u) {
      const int px = min(px0 + 16 * u, CS3_WH * CS3_WW - 1);
      const int hh = min(max(t.ho0 - 1 + px / CS3_WW, 0), t.H - 1), ww = min(max(t.wo0 - 1 + px % CS3_WW, 0), t.W - 1);
      win[u] = *(const half8*)(t.xb + (unsigned)((hh * t.W + ww) * CS3_C + ch8));
    }
  };
  auto win_commit = [&](const ConvTile& t) {                // pixels outside the image become zero on the way to LDS
    int px0 = tid >> 5;
    MQ_PIN(px0);
    px0 &= 15;
#pragma unroll
    for (int u = 0; u < CS3_NU; ++u) {
      const int px = px0 + 16 * u;
      const int hh = t.ho0 - 1 + px / CS3_WW, ww = t.wo0 - 1 + px % CS3_WW;
      const bool inside = ((unsigned)hh < (unsigned)t.H) & ((unsigned)ww < (unsigned)t.W);   // `&`: selects, not branches
      if (px < CS3_WH * CS3_WW) *(half8*)(Win + px * CS3_XP + ch8) = inside ? win[u] : zero8();
    }
  };
#undef MQ_PIN

  ConvTile cur = conv3_tile(p, tile);
  win_issue(cur);
  for (; tile < tend; ++tile) {
    win_commit(cur);
    __syncthreads();
    ConvTile nxt = cur;
    if (tile + 1 < tend) {
      nxt = conv3_tile(p, tile + 1);
      win_issue(nxt);
    }
    float4_ acc[CS3_PH][2];
#pragma unroll
    for (int i = 0; i < CS3_PH; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (float4_){0.f, 0.f, 0.f, 0.f};
    const half_t* a0 = Win + l15 * CS3_XP + wave * 32 + lg * 8;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int h = 0; h < 2; ++h) {                          // four patch rows at a time: 4 A fragments live
        half8 af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const half8*)(a0 + ((4 * h + i + dy) * CS3_WW + dx) * CS3_XP);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[4 * h + i][j] = mfma16(af[i], wr[tap][j], acc[4 * h + i][j]);
      }
    }
    __syncthreads();                                         // every wave is done with the window
    // ---- exchange: C layout of acc[i][j]: row = position 4*lg + r of patch row i, col = channel j*16 + l15
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < CS3_PH; ++i) *(float4_*)(Os + (wave * 32 + j * 16 + l15) * CS3_OP + i * CS3_PW + 4 * lg) = acc[i][j];
    __syncthreads();
    for (int c = tid; c < p.N * CS3_PH * CS3_PW; c += CS3_NT) {
      const int n = c / (CS3_PH * CS3_PW), pos = c % (CS3_PH * CS3_PW);
      const int ho = cur.ho0 + pos / CS3_PW, wo = cur.wo0 + pos % CS3_PW;
      const float* o = Os + n * CS3_OP + pos;
      constexpr int WS = 32 * CS3_OP;
      float s = ((o[0] + o[WS]) + (o[2 * WS] + o[3 * WS])) + ((o[4 * WS] + o[5 * WS]) + (o[6 * WS] + o[7 * WS]));
      if (p.bias) s += (float)p.bias[n];
      if (ho < cur.H && wo < cur.W) cur.ob[((long)n * cur.H + ho) * cur.W + wo] = s;
    }
    __syncthreads();                                         // the exchange buffer is the next window
    cur = nxt;
  }
}

struct mq_conv_level {          // mirrors include/mqdet_hip.h
  const void* x; float* out; long x_bs; int H, W;
};

// levels[i]: x [B,H,W,256] 16-bit NHWC (batch stride x_bs), out [B,N,H,W] fp32 NCHW; w [32, 9*256] 16-bit (k = tap*256 + c, rows >= N zero),
// bias [N] 16-bit or NULL; stride 1, pad 1.  -1: unsupported shape (callers use mq_conv3x3_nchw32_v2_fwd per level).
extern "C" int MQ_SYM(mq_conv3x3_nchw32_group_fwd)(const mq_conv_level* levels, int nl, const void* w, const void* bias, int B, int C, int N,
                                                   void* stream) {
  if (B <= 0 || nl <= 0) return 0;
  if (nl > CS3_MAX_LEVELS || C != CS3_C || N < 1 || N > 32) return -1;
  ConvGroupParams p;
  p.w = (const half_t*)w; p.bias = (const half_t*)bias; p.B = B; p.N = N; p.nl = 0;
  int first = 0;
  for (int i = 0; i < nl; ++i) {
    const mq_conv_level& a = levels[i];
    if (a.H <= 0 || a.W <= 0) continue;
    if ((long)a.H * a.W * CS3_C >= (1L << 31)) return -1;    // 32-bit element offsets inside one image of a level
    ConvLevel& v = p.lv[p.nl++];
    v.x = (const half_t*)a.x; v.out = a.out; v.x_bs = a.x_bs; v.H = a.H; v.W = a.W; v.pad_ = 0;
    v.tiles_y = (a.H + CS3_PH - 1) / CS3_PH; v.tiles_x = (a.W + CS3_PW - 1) / CS3_PW;
    v.first_tile = first;
    first += B * v.tiles_y * v.tiles_x;
  }
  if (p.nl == 0) return 0;
  for (int i = p.nl; i < CS3_MAX_LEVELS; ++i) p.lv[i] = p.lv[0];
  p.tiles_total = first;
  p.tiles_per_xcd = (first + 7) / 8;
  const int per_xcd_blocks = max(1, min(mq_device_cus() / 8, p.tiles_per_xcd));    // one workgroup per CU
  p.tiles_per_block = (p.tiles_per_xcd + per_xcd_blocks - 1) / per_xcd_blocks;
  const int blocks = (p.tiles_per_xcd + p.tiles_per_block - 1) / p.tiles_per_block;   // per XCD, all with >= 1 tile (the last XCD may have fewer)
  constexpr size_t window = (size_t)CS3_WH * CS3_WW * CS3_XP * sizeof(half_t);
  constexpr size_t ostage = (size_t)CS3_NW * 32 * CS3_OP * sizeof(float);
  constexpr size_t smem = window > ostage ? window : ostage;
  static MqOncePerDevice attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set.done();
  }
  hipLaunchKernelGGL(conv3x3_group_kernel, dim3((unsigned)(8 * blocks)), dim3(CS3_NT), smem, (hipStream_t)stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

MQ_NAMESPACE_END
