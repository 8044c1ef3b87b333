// mq_dcnv2_fwd: DCNv2 (modulated deformable 3x3 conv, 256 -> 256 channels) as ONE implicit-GEMM MFMA kernel, gfx950.
//
//   out[m, n] = bias[n] + sum_{tap, c} sigmoid(ml) * bilinear(x[b, :, :, c], ho*s-1+ky+dh, wo*s-1+kx+dw) * W[n, tap*C + c]
//
// Reference: maskrcnn_benchmark/csrc/cuda/deform_conv_kernel_cuda.cu:578-640 (im2col, bilinear :475-503) +
// deform_conv_cuda.cu:538-560 (fp32 column buffer [C*9, Ho*Wo] per sample -> addmm).  No column matrix here: the
// bilinear gather feeds MFMA A-tiles through registers into LDS.  The flat-offset quirk (offsets / mask logits of
// om[b, 27, oH, oW] indexed by the OUTPUT dims) is kept.
//
// v1 (conv_igemm.hip, 4 waves, one k-step of prefetch) was gather-LATENCY bound (187 TFLOP/s); v2 (8 waves, two-slot
// register ring) reached 215 TFLOP/s and was bound by L2 -> CU ingest of the gather (18 KB per output position, ~8 B/clk/CU
// -- the same rate the stand-alone im2col kernel gets).  v3 makes the gather hit the CU's 32 KB vector L1 instead:
//   * a workgroup owns an 8 x 16 PATCH of output positions (not 128 consecutive ones) and walks k as
//     (64-channel slice) outer, (tap) inner: the 9 taps x 4 corners of one slice all land in the same
//     ~(8+2) x (16+2) pixels x 128 B = 23 KB of input, one full cache line per (pixel, slice), ~36-fold reuse out of L1;
//     the weight tile is streamed with non-temporal loads so that it does not evict those lines;
//   * 512 threads = 8 waves (2 per SIMD): tile 128 positions x 256 channels x 64 k, wave grid 2 x 4 (64 x 64 each);
//   * all per-(row, tap) sampling state (4 corner offsets, 4 weights x mask) is computed ONCE into LDS in the prologue,
//     so no global load other than the tile prefetch is ever consumed inside the k-loop (which would make hipcc drain
//     the VMEM queue);
//   * a two-slot register ring keeps the gathers of k-steps ks+1 and ks+2 in flight while step ks runs on the MFMAs
//     (24 x 16-byte loads per thread outstanding), LDS tiles are double-buffered, one barrier per step.
#include "common.h"
#include <type_traits>

struct DcnFParams {
  const half_t* x; const half_t* w; const half_t* bias; const float* om; half_t* out;
  long x_bs;
  int B, H, W, C, Ho, Wo, stride, oH, oW, out_ld, tiles_x, tiles_y, tiles_total;
};

struct alignas(16) TapState { int off[4]; float w[4]; };

static constexpr int DCN_PH = 8, DCN_PW = 16;                // patch of output positions per workgroup

__global__ __launch_bounds__(512) void dcn_igemm8_kernel(DcnFParams p) {
  constexpr int BM = DCN_PH * DCN_PW, BN = 256, BK = 64, LP = BK + 8;
  static_assert(BM == 128, "tile is 128 positions");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* As = (half_t*)smem;                                // [2][BM][LP]
  half_t* Bs = As + 2 * BM * LP;                             // [2][BN][LP]
  TapState* Ts = (TapState*)(Bs + 2 * BN * LP);              // [BM][9]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;
  const int n_pos = p.Ho * p.Wo;
  // XCD-aware tile order: workgroup i runs on XCD i % 8 (each XCD has its own 4 MB L2), so XCD x gets the CONTIGUOUS range
  // of patches [x * tpx, (x + 1) * tpx): the ~36-fold re-use of every input line (9 taps x 4 corners x neighbours) is then
  // served by that XCD's L2 (~50 B/clk/CU) instead of the Infinity Cache / HBM path (~11 B/clk/CU, tools/ingest_microbench).
  const int tpx = (p.tiles_total + 7) >> 3;
  const int tile = (blockIdx.x & 7) * tpx + (blockIdx.x >> 3);
  if (tile >= p.tiles_total) return;
  const int b = tile / (p.tiles_x * p.tiles_y), trem = tile % (p.tiles_x * p.tiles_y);
  const int ho0 = (trem / p.tiles_x) * DCN_PH, wo0 = (trem % p.tiles_x) * DCN_PW;
  const int K = 9 * p.C;
  const int nslice = p.C / BK;
  const int ksteps = 9 * nslice;                             // k-step ks = (slice ks / 9, tap ks % 9)

  // ---- prologue: sampling state of every (row, tap) of this patch -> LDS
  for (int t = tid; t < BM * 9; t += 512) {
    const int row = t / 9, tap = t % 9;
    const int ho = ho0 + row / DCN_PW, wo = wo0 + row % DCN_PW;
    const bool ok_row = ho < p.Ho && wo < p.Wo;
    const int pos = ok_row ? ho * p.Wo + wo : 0;
    const float* omb = p.om + (long)b * 27 * p.oH * p.oW;
    const float dh = omb[(long)(2 * tap) * n_pos + pos];
    const float dw = omb[(long)(2 * tap + 1) * n_pos + pos];
    const float ml = omb[(long)18 * p.oH * p.oW + (long)tap * n_pos + pos];
    const float mk = 1.f / (1.f + __expf(-ml));
    const float hf = (float)(ho * p.stride - 1 + tap / 3) + dh, wf = (float)(wo * p.stride - 1 + tap % 3) + dw;
    const bool inside = ok_row && hf > -1.f && wf > -1.f && hf < (float)p.H && wf < (float)p.W;
    const int h0 = (int)floorf(hf), w0 = (int)floorf(wf);
    const float lh = hf - (float)h0, lw = wf - (float)w0;
    const float wq[4] = {(1.f - lh) * (1.f - lw), (1.f - lh) * lw, lh * (1.f - lw), lh * lw};
    TapState st;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int hh = h0 + (q >> 1), ww = w0 + (q & 1);
      const bool ok = inside && hh >= 0 && hh <= p.H - 1 && ww >= 0 && ww <= p.W - 1;
      st.off[q] = ok ? (hh * p.W + ww) * p.C : 0;            // invalid corners: harmless address, zero weight
      st.w[q] = ok ? wq[q] * mk : 0.f;
    }
    Ts[t] = st;
  }

  // ---- this thread's A tasks: row = tid / 4, 8-channel chunks (tid % 4) and (tid % 4) + 4;  B tasks: chunks tid + j*512
  const int arow = tid >> 2, ach = tid & 3;
  const half_t* xb = p.x + (long)b * p.x_bs;
  __syncthreads();

  float c_w[2][4];
  half8 a_raw[2][2][4], b_raw[4];
  auto issue_a = [&](auto SLOT, int ks) {                    // gather of k-step ks (distance 2)
    constexpr int s = decltype(SLOT)::value;
    const int slice = ks / 9, tap = ks - slice * 9;
    const TapState st = Ts[arow * 9 + tap];
#pragma unroll
    for (int q = 0; q < 4; ++q) c_w[s][q] = st.w[q];
    const int cbase = slice * BK + ach * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) a_raw[s][e][q] = *(const half8*)(xb + st.off[q] + cbase + e * 32);
  };
  auto issue_b = [&](int ks) {                               // weight tile of k-step ks (distance 1: L2-resident, regular)
    const int slice = ks / 9, tap = ks - slice * 9;
    const half_t* wk = p.w + tap * p.C + slice * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + j * 512;
      b_raw[j] = __builtin_nontemporal_load((const half8*)(wk + (long)(c >> 3) * K + (c & 7) * 8));
    }
  };
  auto commit = [&](auto SLOT, int buf) {
    constexpr int s = decltype(SLOT)::value;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += c_w[s][q] * (float)a_raw[s][e][q][j];
      half8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (half_t)acc[j];
      *(half8*)(As + (buf * BM + arow) * LP + ach * 8 + e * 32) = v;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + j * 512;
      *(half8*)(Bs + (buf * BN + (c >> 3)) * LP + (c & 7) * 8) = b_raw[j];
    }
  };

  float4_ acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4_){0.f, 0.f, 0.f, 0.f};

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  issue_a(S0{}, 0);
  issue_b(0);
  issue_a(S1{}, 1);
  commit(S0{}, 0);
  __syncthreads();

  auto body = [&](int ks, auto SLOT, auto OTHER) {
    const int cur = ks & 1;
    // slot of step ks is free again: prefetch step ks + 2 into it.  Unconditional (the last two steps re-load the final
    // tile) so that hipcc's vmcnt bookkeeping sees ONE path and lets these loads stay in flight across the next commit.
    issue_b(min(ks + 1, ksteps - 1));                    // older than the gather below: commit() waits for it with vmcnt(8)
    issue_a(SLOT, min(ks + 2, ksteps - 1));
    __builtin_amdgcn_sched_barrier(0);                   // keep the prefetch ahead of everything that waits on VMEM
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      half8 af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const half8*)(As + (cur * BM + wr * 64 + i * 16 + l15) * LP + kk * 32 + lg * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const half8 bf = *(const half8*)(Bs + (cur * BN + wc * 64 + j * 16 + l15) * LP + kk * 32 + lg * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = mfma16(af[i], bf, acc[i][j]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    commit(OTHER, cur ^ 1);                              // step ks + 1 was issued one full step ago
    __syncthreads();
  };
  for (int ks = 0; ks < ksteps; ks += 2) {               // ksteps = 9 * C/64 is even (C % 128 == 0)
    body(ks, S0{}, S1{});
    body(ks + 1, S1{}, S0{});
  }

  // ---- epilogue: + bias, fp16, transpose through LDS, 16-byte coalesced NHWC stores
  constexpr int OS = BN + 8;
  half_t* Os = (half_t*)smem;                                // [BM][OS] (tiles dead: last barrier passed)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = wc * 64 + j * 16 + l15;
    const float bv = p.bias ? (float)p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) Os[(wr * 64 + i * 16 + lg * 4 + r) * OS + col] = (half_t)(acc[i][j][r] + bv);
  }
  __syncthreads();
  for (int c = tid; c < BM * (BN / 8); c += 512) {
    const int row = c / (BN / 8), ch = c % (BN / 8);
    const int ho = ho0 + row / DCN_PW, wo = wo0 + row % DCN_PW;
    if (ho < p.Ho && wo < p.Wo)
      *(half8*)(p.out + ((long)b * n_pos + ho * p.Wo + wo) * p.out_ld + ch * 8) = *(const half8*)(Os + row * OS + ch * 8);
  }
}

// DCNv2 3x3, pad 1, 256 output channels.  x [B,H,W,C] fp16 NHWC (batch stride x_bs, C % 128 == 0), om [B,27,oH,oW] fp32
// (18 offsets + 9 mask logits, NCHW), w [256, 9*C] fp16 (k = tap*C + c), bias [256] fp16 or NULL, out [B*Ho*Wo, out_ld].
extern "C" int mq_dcnv2_fwd(const void* x, const float* om, const void* w, const void* bias, void* out, int B, int H, int W,
                            int C, long x_bs, int oH, int oW, int N, int out_ld, int stride, void* stream) {
  if (B <= 0) return 0;
  if (N != 256 || C % 128 || stride < 1 || stride > 2 || out_ld < N || out_ld % 8) return -1;
  DcnFParams p;
  p.x = (const half_t*)x; p.w = (const half_t*)w; p.bias = (const half_t*)bias; p.om = om; p.out = (half_t*)out;
  p.x_bs = x_bs; p.B = B; p.H = H; p.W = W; p.C = C; p.stride = stride; p.oH = oH; p.oW = oW; p.out_ld = out_ld;
  p.Ho = (H + 2 - 3) / stride + 1; p.Wo = (W + 2 - 3) / stride + 1;
  if ((long)p.Ho * p.Wo > (long)oH * oW) return -2;          // flat reads must stay inside the om buffer
  p.tiles_y = (p.Ho + DCN_PH - 1) / DCN_PH; p.tiles_x = (p.Wo + DCN_PW - 1) / DCN_PW;
  p.tiles_total = B * p.tiles_y * p.tiles_x;
  constexpr size_t tiles = (size_t)(2 * 128 * 72 + 2 * 256 * 72) * sizeof(half_t) + 128 * 9 * sizeof(TapState);
  constexpr size_t ostage = (size_t)128 * (256 + 8) * sizeof(half_t);
  constexpr size_t smem = tiles > ostage ? tiles : ostage;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)dcn_igemm8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(dcn_igemm8_kernel, dim3((unsigned)(8 * ((p.tiles_total + 7) / 8))), dim3(512), smem, (hipStream_t)stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}
