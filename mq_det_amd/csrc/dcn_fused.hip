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
//   * a two-slot register ring keeps the gathers of two k-steps in flight, LDS tiles are double-buffered;
//   * v5 "ping-pong": the two waves of a SIMD alternate roles every half step -- one runs the MFMAs of step ks while
//     the other does the bilinear blend / LDS staging of step ks + 1 -- so VALU + LDS work hides under the matrix pipe.
#include "common.h"
#include <type_traits>
#include <cstdlib>

MQ_NAMESPACE_BEGIN

struct DcnFParams {
  const half_t* x; const half_t* w; const half_t* bias; const float* om; half_t* out;
  float* stats;            // optional [B, tiles_y*tiles_x, 256, 3]: per-patch (sum y, sum y^2, sum w_p y) of the fp16 output
  const float* wy; const float* wx;    // position weights w_p = wy[ho]*wx[wo] of the third statistic; NULL -> 1/(Ho*Wo)
  long x_bs;
  int B, H, W, C, Ho, Wo, stride, oH, oW, out_ld, tiles_x, tiles_y, tiles_total;
  int mask_prob;           // 1: om[18..26] holds mask PROBABILITIES (the reference operator's argument), 0: logits (sigmoid here)
};

// per (output position, tap): BYTE offsets of the 4 bilinear corners inside the image, and corner weight x mask
struct alignas(16) TapState { unsigned off[4]; float w[4]; };

static constexpr int DCN_PH = 8, DCN_PW = 16;                // patch of output positions per workgroup

// One launch for SEVERAL convolutions ("branches": the up-to-13 DCNv2 calls of one DyConv layer, vldyhead.py:205-247).
// A branch alone gives 8 ... 1144 tiles at B = 8 and every launch rounds its tile count up to whole waves of 256 CUs
// (263 tiles -> two rounds, the second with 7 workgroups); 13 launches on 5 streams cost ~2.7x the CU time of their
// tiles (profiles/README.md "launch quantisation").  Grouped, the ~2100 tiles of a layer fill the chip back to back.
static constexpr int DCN_MAX_BRANCH = 16;
struct DcnGroup {
  DcnFParams br[DCN_MAX_BRANCH];
  int first_tile[DCN_MAX_BRANCH + 1];                        // prefix sums of tiles_total
  int n, tiles_all;
};

// NW = waves per workgroup: 8 (two per SIMD, wave tile 64 x 64) or 16 (four per SIMD, wave tile 32 x 64, <= 128 VGPRs): the
// staging phase is a dependent LDS -> VALU -> LDS chain, more resident waves overlap more of those chains.
// ABL (tools/microbench.py only; results are garbage): the kernel WITHOUT one of its parts -- bit 0: no gather loads, bit 1: no bilinear
// blend (corner 0 is staged as it is), bit 2: no weight-tile loads, bit 3: no fragment reads / MFMAs -- to see what a k-step waits for.
// SYNC = barriers per k-step.  2: one after each half step (round 1 .. 3: the two wave groups swap roles in lock step).  1: only the one
// that ends a step.  Within a step group 0 runs [MFMAs of step k, then its share of the staging of step k + 1] and group 1 the same two
// in the opposite order: the MFMAs only READ buffer k & 1, the staging only WRITES buffer (k + 1) & 1, each thread its own rows -- nothing
// inside a step depends on the other group, only the step boundary does (everybody's share of step k + 1 must be visible before its
// MFMAs, everybody must be done reading buffer k & 1 before step k + 2 is staged into it).  The ablation of round 3 (GPU call 10) put
// the kernel's skeleton -- mostly its 72 barriers of 1024 threads per tile -- at a third of the launch.
// PLAIN (round 5): every branch of the launch is a PLAIN 3 x 3 convolution handed over as a DCNv2 with all-zero offsets and mask 1 (the FPN
// output convs, pipeline.fpn_forward; flags bit 1 of mq_dcn_branch is the caller's promise).  A tap then samples ONE pixel: corner 0 with
// weight 1 (0 where the tap lies in the zero padding), corners 1 .. 3 with weight exactly 0 -- so only corner 0 is gathered and the blend is
// one multiply by 0 / 1.  Same results bit for bit as the general path (w0 a + 0 b + 0 c + 0 d in the same fma order), a quarter of the gather
// loads and none of the blend arithmetic.
// FENCE (round 5, a NEGATIVE result kept as a template switch, not instantiated by the launcher): a scheduling fence behind every barrier of the
// k-loop.  A workgroup barrier orders LDS traffic, not arithmetic: hipcc hoists the bilinear blend of the NEXT half step (registers only) above
// the barrier, and with it the wait for that step's gather -- the shipped ISA drains its loads with vmcnt(0) once per two k-steps.  Pinning the
// order (waits bottom out at vmcnt(4), as the source intends) made the launch 9 % SLOWER (3.72 vs 3.40 ms per step, three of three, GPU call 13 of
// round 5, profiles/r05_call13_dcn_fence_ab.txt; equal outputs): the hoisted blend fills the issue slots beside the other wave group's MFMAs, and the
// gather has landed by then anyway (one k-step is ~1.7 us).
// SPLIT-PRECISE build (-DMQ_F32, round 6): the SAME byte geometry with 4-byte operands -- a k-step is 32 channels (128 B per row and corner in
// global memory, 16-byte chunks of 4 elements), and the LDS tiles are PLANAR: every operand element is split ONCE, by the thread that stages it,
// into hi = fp16(x) and lo = fp16((x - hi) 2^11) (csrc/common.h) stored in two fp16 planes [rows][32]; a fragment read is one ds_read_b128 per
// plane and the MFMA phase is three v_mfma_f32_16x16x32_f16 per (row block, column block) with no conversion arithmetic next to them.  Planes
// have a 64-byte pitch: the 16 rows x 4 chunks of a fragment read are 1 KB of consecutive bytes (no swizzle needed).  Tiles stay double-buffered
// (2 x 48 KB) and the ping-pong of the two wave groups is the one of the 16-bit build.  (Round 5 ran this build with fp32 tiles, ONE buffer,
// eight v_mfma_f32_16x16x4_f32 per MFMA and 117 spilled VGPRs: 6.4 ms per launch against 0.56 ms with fp16 operands.)
#if defined(MQ_F32) && !defined(MQ_F32_EXACT)
#define MQ_DCN_SPLIT 1
#else
#define MQ_DCN_SPLIT 0
#endif
#if MQ_DCN_SPLIT
typedef mq_f32x4 dcn_gvec;                                   // one 16-byte chunk of operand elements in global memory
#else
typedef half8 dcn_gvec;
#endif
// BDMA (round 6; flags bit 2 of every branch / KERNELS["DCN_BDMA"], the default): the weights arrive in LDS-TILE ORDER -- per k-step one 32 KB block
// that is the byte image of the B tile (16-bit builds: rows of 64 with the chunk swizzle applied; split-precise: [hi plane | lo plane] of rows of
// 32; mq_det_amd.ops.dcn_weight_tiles) -- and are copied global -> LDS by LDS-DMA: no weight registers, no ds_write, no split arithmetic for B.
// One barrier per k-step.  Group 0: MFMAs(k), blend + staging of A(k + 1), gathers of A(k + 3) -- no copy, plain barriers, counted waits by hipcc.
// Group 1: blend + staging of A(k + 1), copy of B(k + 1), gathers of A(k + 3), MFMAs(k), then vmcnt(<its gathers>) = "the copy has landed".
// Three things the compiler's wait-count pass needed (each one was a vmcnt(0) in the middle of the step; GPU calls 12 / 13: 0.603 -> 0.515 ms):
//   * the copy as MUBUF `buffer_load ... lds`: the FLAT-encoded global_load_lds is booked as a flat access ("pending flat": every later wait of
//     the loop becomes vmcnt(0) / lgkmcnt(0));
//   * group 1's loop behind __restrict__ tile pointers (dcn_scoped_tiles): without alias scopes every LDS read after a copy waits for the copy;
//   * the end-of-step wait as __builtin_amdgcn_s_waitcnt, which the pass books, not as an asm string.
// dynamic LDS of the kernel without the row table behind it: max(tiles + sampling state, O staging + statistics partials)
constexpr size_t dcn_smem_main() {
  constexpr size_t nbuf = (sizeof(half_t) == 4 && !MQ_DCN_SPLIT) ? 1 : 2;
  constexpr size_t bk = MQ_DCN_SPLIT ? 32 : 64;
  constexpr size_t tiles = (size_t)(nbuf * 128 * bk + nbuf * 256 * bk) * sizeof(half_t) + 128 * 9 * sizeof(TapState);
  constexpr size_t ostage = (size_t)128 * (256 + 8) * sizeof(half_t) + (sizeof(half_t) == 4 ? 0 : (size_t)16 * 32 * 24 * sizeof(float));
  return tiles > ostage ? tiles : ostage;
}
// the tiles of the two buffers and the sampling state as DISJOINT objects (alias scopes once inlined)
template <class T, class S, class F>
__device__ __forceinline__ void dcn_scoped_tiles(T* __restrict__ a0, T* __restrict__ a1, T* __restrict__ b0, T* __restrict__ b1, const S* __restrict__ ts, F f) {
  f(a0, a1, b0, b1, ts);
}

template <int NW, int ABL = 0, int SYNC = 2, bool PLAIN = false, bool FENCE = false, bool BDMA = false>
__global__ __launch_bounds__(64 * NW) void dcn_igemm8_kernel(DcnGroup g) {
  constexpr int BM = DCN_PH * DCN_PW, BN = 256, BK = MQ_DCN_SPLIT ? 32 : 64;
  constexpr int CH = 16 / (int)sizeof(half_t);               // operand elements per 16-byte chunk (8, or 4 in the split-precise build)
  constexpr int NTH = 64 * NW, RA = 1024 / NTH, JB = 2048 / NTH, IM = 32 / NW;   // threads, A rows / thread, B chunks / thread, row blocks / wave
  static_assert(BM == 128, "tile is 128 positions");
  static_assert(BK * sizeof(half_t) == 128, "a tile row is one 128-byte line per corner");
  // tile buffers: two (ping-pong); only the exact-fp32 build (-DMQ_F32_EXACT: fp32 tiles of 64 channels) has ONE, where every wave runs
  // MFMAs -> barrier -> staging -> barrier
  constexpr int NBUF = (sizeof(half_t) == 4 && !MQ_DCN_SPLIT) ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS tiles: rows of 64 halfs = 128 B = 8 chunks of 16 B, NO padding; chunk c of row r is stored at chunk position
  // c ^ (r & 7): conflict-free for the ds_read_b128 fragment reads (16 rows x one chunk) AND for the row-wise
  // ds_write_b128 of the staging pass (8 lanes = one row).  (The padded 144-byte pitch of v3 lost 39 % of the LDS
  // cycles to bank conflicts, profiles/r01_pmc_dcn_v3.txt.)
  half_t* As = (half_t*)smem;                                // [2][BM][BK]   (split-precise: [2][hi | lo][BM][32] fp16 -- the same bytes)
  half_t* Bs = As + NBUF * BM * BK;                          // [NBUF][BN][BK]
  TapState* Ts = (TapState*)(Bs + NBUF * BN * BK);           // [BM][9]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  // XCD-aware tile order: workgroup i runs on XCD i % 8 (each XCD has its own 4 MB L2), so XCD x gets the CONTIGUOUS range
  // of patches [x * tpx, (x + 1) * tpx): the ~36-fold re-use of every input line (9 taps x 4 corners x neighbours) is then
  // served by that XCD's L2 (~50 B/clk/CU) instead of the Infinity Cache / HBM path (~11 B/clk/CU, tools/ingest_microbench).
  // (grouped launch: the same mapping over the concatenated tile list; tiles of one branch / image stay contiguous)
  const int tpx = (g.tiles_all + 7) >> 3;
  int tile = (blockIdx.x & 7) * tpx + (blockIdx.x >> 3);
  if (tile >= g.tiles_all) return;
  int bi = 0;
  while (bi + 1 < g.n && tile >= g.first_tile[bi + 1]) ++bi;           // <= 13 branches: scalar scan
  tile -= g.first_tile[bi];
  const DcnFParams p = g.br[bi];
  const int n_pos = p.Ho * p.Wo;
  // BANDED tiles (tiles_y == 0, the default since GPU call 20 of round 6).  8 x 16 patches (MQ_DCN_RASTER=0) round BOTH extents of a level up:
  // 13 x 11 patches for the 100 x 168 level (9 % empty rows), 7 x 6 for 50 x 84 (28 %), 4 x 3 for 25 x 42 (46 %) -- 15 % of the rows of a DyConv
  // layer's launch at the bench shape were computed and thrown away.  Now the output positions of an image are ORDERED band by band (bands of 8
  // rows; the last band has Ho % 8), inside a band column by column, and tile t = positions [128 t, 128 t + 128) of that order: a tile is still an
  // 8-row x 16-column patch (or the end of one band + the start of the next), so the input footprint and its L2 / L1 reuse stay those of a patch,
  // and only the last tile of an image is partly empty.  (Plain row-major tiles -- GPU call 18 -- have no empty rows either, but a 128-position
  // strip of one row touches 2 x the input lines of a patch: 7 % MORE time per k-step, the launch no faster.)
  const bool raster = p.tiles_y == 0;
  const int tpi = raster ? p.tiles_x : p.tiles_x * p.tiles_y;
  const int b = tile / tpi, trem = tile % tpi;
  const int ho0 = raster ? 0 : (trem / p.tiles_x) * DCN_PH, wo0 = raster ? 0 : (trem % p.tiles_x) * DCN_PW;
  const int p0 = trem * BM;                                  // banded: first position (in band order) of the tile
  const int full = (p.Ho / DCN_PH) * DCN_PH * p.Wo;          // positions in the full bands
  const int hlast = max(p.Ho % DCN_PH, 1);                   // rows of the last, partial band
  // band order: position q -> (ho, wo)
  auto band_pos = [&](int q, int& ho, int& wo) {
    if (q < full) {
      const int band = q / (DCN_PH * p.Wo), r = q - band * (DCN_PH * p.Wo);
      ho = band * DCN_PH + (r & (DCN_PH - 1)); wo = r / DCN_PH;
    } else {
      const int r = q - full;
      wo = r / hlast; ho = (p.Ho / DCN_PH) * DCN_PH + (r - wo * hlast);
    }
  };
  // the tile's 128 rows -> (ho << 16 | wo << 1 | inside), computed once (two runtime divisions per row; every later use is one LDS read)
  int* rowtab = (int*)(smem + dcn_smem_main());
  if (raster) {
    if (tid < BM) {
      int ho, wo;
      band_pos(min(p0 + tid, n_pos - 1), ho, wo);
      rowtab[tid] = (ho << 16) | (wo << 1) | (p0 + tid < n_pos ? 1 : 0);
    }
    __syncthreads();
  }
  // (ho, wo) of tile row `row`; false: the row is outside the image
  auto row_pos = [&](int row, int& ho, int& wo) -> bool {
    if (raster) {
      const int v = rowtab[row];
      ho = v >> 16; wo = (v >> 1) & 0x7fff;
      return v & 1;
    }
    ho = ho0 + row / DCN_PW; wo = wo0 + row % DCN_PW;
    return ho < p.Ho && wo < p.Wo;
  };
  const int K = 9 * p.C;
  const int nslice = p.C / BK;
  const int ksteps = 9 * nslice;                             // k-step ks = (slice ks / 9, tap ks % 9)
  // ---- L2 warm-up.  Every step of the k-loop gathers lines this XCD has never touched (compulsory misses), and a
  // wave's wait ends with its slowest lane: measured, the step time was the loaded HBM latency (~1.5 us) whatever else
  // the step did (profiles/README.md, DCN section).  So the expected footprint of the patch -- the regular 3x3 window
  // grown by DCN_WARM pixels, all channel lines -- is touched once here, overlapped with the sampling-state prologue;
  // the loop's gathers then find their lines in the L2.  The loads are inline asm (a C++ load without a consumer is
  // dropped, a volatile one is waited for on the spot); their destination registers stay reserved until the first
  // counted wait of the k-loop prologue has passed (VMEM returns in order, so they have landed by then).
  constexpr int DCN_WARM = 1, WARM_N = 4096 / NTH;
  unsigned warm[WARM_N];
  {
    // footprint: one rectangle for a patch; banded tiles: the piece in the first band [+ the piece in the last band, if the tile crosses]
    int fh = (DCN_PH - 1) * p.stride + 3 + 2 * DCN_WARM, fw = (DCN_PW - 1) * p.stride + 3 + 2 * DCN_WARM;
    int h_lo = ho0 * p.stride - 1 - DCN_WARM, w_lo = wo0 * p.stride - 1 - DCN_WARM;
    int fh2 = 0, fw2 = 1, h_lo2 = 0, w_lo2 = 0;
    if (raster) {
      int ha, wa, hb, wb;
      row_pos(0, ha, wa);
      row_pos(min(BM, n_pos - p0) - 1, hb, wb);
      const int ba = ha / DCN_PH, bb = hb / DCN_PH;                        // bands of the first / last position
      const int ra = min(DCN_PH, p.Ho - ba * DCN_PH), rb = min(DCN_PH, p.Ho - bb * DCN_PH);
      h_lo = ba * DCN_PH * p.stride - 1 - DCN_WARM; fh = (ra - 1) * p.stride + 3 + 2 * DCN_WARM;
      w_lo = wa * p.stride - 1 - DCN_WARM;
      fw = ((ba == bb ? wb : p.Wo - 1) - wa) * p.stride + 3 + 2 * DCN_WARM;
      if (ba != bb) {
        h_lo2 = bb * DCN_PH * p.stride - 1 - DCN_WARM; fh2 = (rb - 1) * p.stride + 3 + 2 * DCN_WARM;
        w_lo2 = -1 - DCN_WARM; fw2 = wb * p.stride + 3 + 2 * DCN_WARM;
      }
    }
    const int lpp = (p.C * (int)sizeof(half_t)) >> 7;        // 128-byte lines per pixel
    const int nl1 = fh * fw * lpp, nlines = nl1 + fh2 * fw2 * lpp;
    const char* xw = (const char*)(p.x + (long)b * p.x_bs);
#pragma unroll
    for (int i = 0; i < WARM_N; ++i) {
      warm[i] = 0;
      const int idx = tid + i * NTH;
      if (idx < nlines) {
        const bool second = idx >= nl1;
        const int id2 = second ? idx - nl1 : idx, fwx = second ? fw2 : fw;
        const int px = id2 / lpp, ln = id2 - px * lpp;
        const int hh = min(max((second ? h_lo2 : h_lo) + px / fwx, 0), p.H - 1), ww = min(max((second ? w_lo2 : w_lo) + px % fwx, 0), p.W - 1);
        const char* a = xw + ((long)(hh * p.W + ww) * p.C * (long)sizeof(half_t) + ln * 128);
        asm volatile("global_load_dword %0, %1, off" : "=v"(warm[i]) : "v"(a) : "memory");
      }
    }
  }

  // ---- prologue: sampling state of every (row, tap) of this patch -> LDS.  All offset / mask loads of a thread's (up
  // to 3) tasks are issued before any of them is consumed (one memory round trip instead of three).
  {
    constexpr int NTASK = (BM * 9 + NTH - 1) / NTH;
    float dh[NTASK], dw[NTASK], ml[NTASK];
    const float* omb = p.om + (long)b * 27 * p.oH * p.oW;
#pragma unroll
    for (int i = 0; i < NTASK; ++i) {
      const int t = min(tid + i * NTH, BM * 9 - 1);
      const int row = t / 9, tap = t - row * 9;
      int ho, wo;
      const int pos = row_pos(row, ho, wo) ? ho * p.Wo + wo : 0;
      dh[i] = omb[(long)(2 * tap) * n_pos + pos];
      dw[i] = omb[(long)(2 * tap + 1) * n_pos + pos];
      ml[i] = omb[(long)18 * p.oH * p.oW + (long)tap * n_pos + pos];
    }
#pragma unroll
    for (int i = 0; i < NTASK; ++i) {
      const int t = tid + i * NTH;
      if (t < BM * 9) {
        const int row = t / 9, tap = t - row * 9;
        int ho, wo;
        const bool ok_row = row_pos(row, ho, wo);
        const float mk = p.mask_prob ? ml[i] : 1.f / (1.f + __expf(-ml[i]));
        const float hf = (float)(ho * p.stride - 1 + tap / 3) + dh[i], wf = (float)(wo * p.stride - 1 + tap % 3) + dw[i];
        const bool inside = ok_row && hf > -1.f && wf > -1.f && hf < (float)p.H && wf < (float)p.W;
        const int h0 = (int)floorf(hf), w0 = (int)floorf(wf);
        const float lh = hf - (float)h0, lw = wf - (float)w0;
        const float wq[4] = {(1.f - lh) * (1.f - lw), (1.f - lh) * lw, lh * (1.f - lw), lh * lw};
        TapState st;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int hh = h0 + (q >> 1), ww = w0 + (q & 1);
          const bool ok = inside && hh >= 0 && hh <= p.H - 1 && ww >= 0 && ww <= p.W - 1;
          st.off[q] = ok ? (unsigned)((hh * p.W + ww) * p.C) * (unsigned)sizeof(half_t) : 0u;  // invalid corners: harmless address, zero weight
          st.w[q] = ok ? wq[q] * mk : 0.f;
        }
        Ts[t] = st;
      }
    }
  }
  __syncthreads();

  // ---- staging tasks.  A: rows ar and ar + 64, 16-byte chunk ac (8 lanes = one 128-byte line of one corner);
  //      B: chunks tid + j*512 (row = chunk / 8).  Addresses are 32-bit byte offsets from wave-uniform bases.
  const int wr = wave >> 2, wc = wave & 3;                   // wave tile: rows wr * IM*16.., columns wc * 64..
  const int ar = tid >> 3, ac = tid & 7;                     // A rows ar + rr * (NTH / 8), rr < RA
  const char* xb = (const char*)(p.x + (long)b * p.x_bs);
  const char* wb = (const char*)p.w;
#if MQ_DCN_SPLIT
  const unsigned a_lds = (unsigned)(ar * BK + ac * CH);                                // fp16 index inside a plane (no swizzle)
#else
  const unsigned a_lds = (unsigned)(ar * BK + ((ac ^ (ar & 7)) << 3));                 // + 64 * BK for the second row
#endif
  unsigned b_goff[JB], b_lds[JB];
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const int c = tid + j * NTH, row = c >> 3, ch = c & 7;
    b_goff[j] = (unsigned)(row * K + ch * CH) * (unsigned)sizeof(half_t);
#if MQ_DCN_SPLIT
    b_lds[j] = (unsigned)(row * BK + ch * CH);
#else
    b_lds[j] = (unsigned)(row * BK + ((ch ^ (row & 7)) << 3));
#endif
  }

  float c_w[2][RA][4];
  dcn_gvec a_raw[2][RA][4], b_raw[JB];
  if constexpr (ABL != 0) {
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int rr = 0; rr < RA; ++rr)
#pragma unroll
        for (int q = 0; q < 4; ++q) a_raw[s_][rr][q] = dcn_gvec{};
#pragma unroll
    for (int j = 0; j < JB; ++j) b_raw[j] = dcn_gvec{};
  }
  auto issue_a_at = [&](auto SLOT, int ks, const TapState* Ts) {   // gather of k-step ks
    constexpr int s = decltype(SLOT)::value;
    ks = min(ks, ksteps - 1);                                // tail: re-load the last step (one code path, no branches)
    const int slice = ks / 9, tap = ks - slice * 9;
    const unsigned cb = (unsigned)(slice * BK + ac * CH) * (unsigned)sizeof(half_t);
#pragma unroll
    for (int rr = 0; rr < RA; ++rr) {
      const TapState st = Ts[(ar + rr * (NTH / 8)) * 9 + tap];
#pragma unroll
      for (int q = 0; q < (PLAIN ? 1 : 4); ++q) {
        c_w[s][rr][q] = st.w[q];
        if constexpr (!(ABL & 1)) a_raw[s][rr][q] = *(const dcn_gvec*)(xb + (st.off[q] + cb));
      }
    }
  };
  auto issue_a = [&](auto SLOT, int ks) { issue_a_at(SLOT, ks, Ts); };
  auto issue_b = [&](int ks) {                               // weight tile of k-step ks
    ks = min(ks, ksteps - 1);
    const int slice = ks / 9, tap = ks - slice * 9;
    const unsigned kb = (unsigned)(tap * p.C + slice * BK) * (unsigned)sizeof(half_t);
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      if constexpr (!(ABL & 4)) b_raw[j] = *(const dcn_gvec*)(wb + (b_goff[j] + kb));
    }
  };
  // staging of one k-step by this thread: bilinear blend of its gathered slot (fp32 accumulate like the im2col kernel,
  // one rounding to fp16) -> its two A-tile rows, and its four weight chunks -> B tile
  auto stage_at = [&](auto SLOT, half_t* a_tile, half_t* b_tile) {      // the two tiles of ONE buffer
    constexpr int s = decltype(SLOT)::value;
#if MQ_DCN_SPLIT
    // planes of the buffer: A hi, A lo, B hi, B lo (fp16); this thread's CH = 4 blended values / weights -> 8 bytes into each plane
    _Float16* a_hi = (_Float16*)a_tile + a_lds;
    _Float16* b_hi = (_Float16*)b_tile;
#pragma unroll
    for (int rr = 0; rr < RA; ++rr) {
      mq_f32x4 t;
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        t[j] = c_w[s][rr][0] * a_raw[s][rr][0][j];
        if constexpr (!PLAIN) {
          t[j] = __builtin_fmaf(c_w[s][rr][1], a_raw[s][rr][1][j], t[j]);
          t[j] = __builtin_fmaf(c_w[s][rr][2], a_raw[s][rr][2][j], t[j]);
          t[j] = __builtin_fmaf(c_w[s][rr][3], a_raw[s][rr][3][j], t[j]);
        }
      }
      mq_h16x4 hi, lo;
      mq_split4(t, hi, lo);
      *(mq_h16x4*)(a_hi + rr * (NTH / 8) * BK) = hi;
      *(mq_h16x4*)(a_hi + BM * BK + rr * (NTH / 8) * BK) = lo;
    }
    if constexpr (!BDMA) {
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        mq_h16x4 hi, lo;
        mq_split4(b_raw[j], hi, lo);
        *(mq_h16x4*)(b_hi + b_lds[j]) = hi;
        *(mq_h16x4*)(b_hi + BN * BK + b_lds[j]) = lo;
      }
    }
#else
    half_t* a = a_tile + a_lds;
#pragma unroll
    for (int rr = 0; rr < RA; ++rr) {
      half8 v;
      if constexpr (ABL & 2) v = a_raw[s][rr][0];
      else
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = c_w[s][rr][0] * (float)a_raw[s][rr][0][j];
        if constexpr (!PLAIN) {
          t = __builtin_fmaf(c_w[s][rr][1], (float)a_raw[s][rr][1][j], t);
          t = __builtin_fmaf(c_w[s][rr][2], (float)a_raw[s][rr][2][j], t);
          t = __builtin_fmaf(c_w[s][rr][3], (float)a_raw[s][rr][3][j], t);
        }
        v[j] = (half_t)t;
      }
      *(half8*)(a + rr * (NTH / 8) * BK) = v;
    }
    if constexpr (!BDMA) {
#pragma unroll
      for (int j = 0; j < JB; ++j) *(half8*)(b_tile + b_lds[j]) = b_raw[j];
    }
#endif
  };
  auto stage = [&](auto SLOT, int buf) {
    buf &= NBUF - 1;
    stage_at(SLOT, As + buf * BM * BK, Bs + buf * BN * BK);
  };
  // BDMA: the B tile of k-step ks (a 32 KB block of the tile-ordered weights) -> one B buffer, issued by the waves of GROUP 1 only (2 JB linear
  // 1 KB pieces per wave); group 0 never has a copy in flight.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // The copy is a MUBUF `buffer_load_dwordx4 ... lds`, not `global_load_lds`: hipcc's wait-count pass books the FLAT-encoded form as a flat access
  // that may touch LDS and memory ("pending flat"), and from then on EVERY s_waitcnt of the loop is vmcnt(0) / lgkmcnt(0) -- no counted waits for
  // the gathers, no ds_read / MFMA overlap inside a wave.  Weights as a raw buffer (base = w, no stride); lane offset in one VGPR, tile and piece
  // offsets in the scalar offset.
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, 0x7fffffff, 0x00020000);
  const unsigned dma_voff = (unsigned)((tid & (NTH / 2 - 1)) * 16);
  auto dma_b_at = [&](int ks, half_t* b_tile) {
    constexpr int TILE_B = BN * BK * (int)sizeof(half_t);     // 32 KB in every build
    constexpr int HT = NTH / 2;                                // threads of group 1
    ks = min(ks, ksteps - 1);
    const int soff = __builtin_amdgcn_readfirstlane(ks * TILE_B);
    char* dst = (char*)b_tile + (wave_u - NW / 2) * 1024;
#pragma unroll
    for (int j = 0; j < 2 * JB; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(dst + j * HT * 16), 16, dma_voff, soff + j * HT * 16, 0, 0);
  };
  auto dma_b = [&](int ks, int buf) { dma_b_at(ks, Bs + (buf & (NBUF - 1)) * BN * BK); };

  float4_ acc[IM][4];
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4_){0.f, 0.f, 0.f, 0.f};

  // fragment addressing: row = w? * 64 + i * 16 + l15 (row & 7 == l15 & 7), chunk = kk * 4 + lg
  const unsigned fa = (unsigned)((wr * IM * 16 + l15) * BK), fb = (unsigned)((wc * 64 + l15) * BK);
#if MQ_DCN_SPLIT
  auto mfma_at = [&](const half_t* a_tile, const half_t* b_tile) {   // one 32-deep step: IM x 4 blocks, three fp16 MFMAs each on the planar fragments
    if constexpr (ABL & 8) return;
    const _Float16* At = (const _Float16*)a_tile + fa + lg * 8;
    const _Float16* Bt = (const _Float16*)b_tile + fb + lg * 8;
    mq_split8 af[IM], bf[4];
#pragma unroll
    for (int i = 0; i < IM; ++i) {
      af[i].hi = *(const mq_h16x8*)(At + i * 16 * BK);
      af[i].lo = *(const mq_h16x8*)(At + BM * BK + i * 16 * BK);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf[j].hi = *(const mq_h16x8*)(Bt + j * 16 * BK);
      bf[j].lo = *(const mq_h16x8*)(Bt + BN * BK + j * 16 * BK);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < IM; ++i) acc[i][j] = mfma16_split(af[i], bf[j], acc[i][j]);
  };
#else
  const unsigned sw0 = (unsigned)((lg ^ (l15 & 7)) << 3), sw1 = (unsigned)(((4 + lg) ^ (l15 & 7)) << 3);
  auto mfma_at = [&](const half_t* a_tile, const half_t* b_tile) {   // this wave's 64 x 64 block of one k-step (32 MFMAs)
    if constexpr (ABL & 8) return;
    const half_t* At = a_tile + fa;
    const half_t* Bt = b_tile + fb;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      half8 af[IM], bf[4];
#pragma unroll
      for (int i = 0; i < IM; ++i) af[i] = *(const half8*)(At + i * 16 * BK + (kk ? sw1 : sw0));
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *(const half8*)(Bt + j * 16 * BK + (kk ? sw1 : sw0));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < IM; ++i) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
    }
  };
#endif
  auto mfma_phase = [&](int cur) {
    cur &= NBUF - 1;
    mfma_at(As + cur * BM * BK, Bs + cur * BN * BK);
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if constexpr (BDMA) {
    // ---- fill: B(0) by DMA (group 1), A(0) gathered and staged by everybody, the gathers of A(1) and A(2) in flight
    issue_a(S0{}, 0);
    stage(S0{}, 0);
    if (wave >= NW / 2) dma_b(0, 0);
    issue_a(S1{}, 1);                                        // slot s holds the tiles of its parity: consumed in step k, refilled with tile k + 3 right after
    issue_a(S0{}, 2);
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0) as an instruction hipcc's wait-count pass SEES: nothing is pending after it
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WARM_N; ++i) asm volatile("" ::"v"(warm[i]));     // warm-up destinations released here
    // step ks reads buffer ks & 1 and fills the other one: group 0 = MFMAs(ks), blend + staging of A(ks + 1), gathers of A(ks + 3);
    // group 1 = blend + staging of A(ks + 1), copy of B(ks + 1), gathers of A(ks + 3), MFMAs(ks); end of step: the copy has landed (it had the
    // MFMA phase to do so), barrier
    // End of a step of group 1: a COUNTED wait.  The step's copies are issued before its gathers (pinned by a sched_barrier;
    // tests/test_host_cpu.py checks the order in the ISA), VMEM returns in order, so vmcnt(<gathers per step>) says "the copies have landed" and
    // leaves the gathers of tile ks + 3 in flight across the barrier (vmcnt(0) here: 0.553 instead of 0.515 ms per launch, GPU call 13).  The
    // wait is the BUILTIN, an instruction hipcc's wait-count pass sees and books (an asm string is invisible to it: it then adds its own
    // vmcnt(0) in front of the barrier's fence and of the next blend).
    constexpr int NG = RA * (PLAIN ? 1 : 4);                 // gather loads per thread and step
    auto step_end = [&]() {
      __builtin_amdgcn_s_waitcnt(0x0F70 | NG);               // vmcnt(NG), nothing else
      __syncthreads();
      // the next step's blend stays below the barrier: hoisted above it, its register wait lands where a copy is still pending
      __builtin_amdgcn_sched_barrier(0);
    };
    if (wave < NW / 2) {
      // group 0 never has a copy in flight: plain barriers, its gathers stay in flight across them (counted waits by hipcc, two steps of distance)
      for (int ks = 0; ks < ksteps; ks += 2) {
        mfma_phase(0); stage(S1{}, 1); issue_a(S1{}, ks + 3);
        __syncthreads();
        mfma_phase(1); stage(S0{}, 0); issue_a(S0{}, ks + 4);
        __syncthreads();
      }
    } else {
      // group 1 runs its loop behind __restrict__ parameters (dcn_scoped_tiles): the accesses carry alias scopes, and hipcc's wait-count pass
      // uses them -- WITHOUT, every LDS read after a copy was issued waits with vmcnt(0) for it (the B tile the MFMAs read and the B tile the
      // copy fills are "the same array" to it): the copy's whole latency in front of the MFMA phase, 9 % slower than register-staged weights
      dcn_scoped_tiles(As, As + BM * BK, Bs, Bs + BN * BK, Ts,
                       [&](half_t* a0, half_t* a1, half_t* b0, half_t* b1, const TapState* ts) __attribute__((always_inline)) {
        for (int ks = 0; ks < ksteps; ks += 2) {
          stage_at(S1{}, a1, b1); dma_b_at(ks + 1, b1);
          __builtin_amdgcn_sched_barrier(0);                  // the copy goes out HERE (hipcc sinks it below the MFMA phase otherwise)
          issue_a_at(S1{}, ks + 3, ts); mfma_at(a0, b0);
          step_end();
          stage_at(S0{}, a0, b0); dma_b_at(ks + 2, b0);
          __builtin_amdgcn_sched_barrier(0);                  // the copy goes out HERE (hipcc sinks it below the MFMA phase otherwise)
          issue_a_at(S0{}, ks + 4, ts); mfma_at(a1, b1);
          step_end();
        }
      });
    }
  } else {
  // ---- pipeline fill: every thread stages step 0; gathers of steps 1, 2 and the weights of step 1 are in flight
  issue_a(S0{}, 0);
  issue_b(0);
  issue_a(S1{}, 1);
  stage(S0{}, 0);
  issue_b(1);
  issue_a(S0{}, 2);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < WARM_N; ++i) asm volatile("" ::"v"(warm[i]));     // warm-up destinations released here

  // ---- k-loop, "ping-pong": the two waves that share a SIMD (wave w and w + 4) are never in the same phase.  While
  // group 0 (waves 0-3) runs its 32 MFMAs of step ks, group 1 stages ITS share of step ks + 1 (blend, LDS stores, next
  // prefetch); then they swap.  (v3/v4: all 8 waves MFMA, then all 8 blend -- 51 % issue stalls,
  // profiles/r01_pmc_dcn_v3.txt.)  Per thread, VMEM issue order is  ... A(k+1) | B(k+2) A(k+3) | ...  so the staging
  // of step k+1 waits with a counted vmcnt: its own gather (issued two steps ago) and weights (one step ago) have
  // landed, the newest gather stays in flight.
  auto stage_next = [&](auto SLOT, int ks) {                 // SLOT holds step ks + 1
    stage(SLOT, (ks + 1) & 1);
    issue_b(ks + 2);
    issue_a(SLOT, ks + 3);
  };
  // One loop per wave group (the branch is wave-uniform; s_barrier only counts arrivals, and both groups execute the
  // same number of barriers).  A single loop with `if (grp == ...)` around the phases makes hipcc merge the two paths'
  // VMEM bookkeeping and wait with vmcnt(0), which throws away one step of prefetch distance.
  if constexpr (NBUF == 1) {
    for (int ks = 0; ks < ksteps; ks += 2) {
      mfma_phase(0);
      __syncthreads();                                       // everybody has read step ks before step ks + 1 overwrites the one buffer
      stage_next(S1{}, ks);
      __syncthreads();
      mfma_phase(1);
      __syncthreads();
      stage_next(S0{}, ks + 1);
      __syncthreads();
    }
  } else if (wave < NW / 2) {
    for (int ks = 0; ks < ksteps; ks += 2) {                 // ksteps = 9 * C/64 is even (C % 128 == 0)
      mfma_phase(0);
      if constexpr (SYNC == 2) __syncthreads();
      stage_next(S1{}, ks);
      __syncthreads();
      if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
      mfma_phase(1);
      if constexpr (SYNC == 2) __syncthreads();
      stage_next(S0{}, ks + 1);
      __syncthreads();
      if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    for (int ks = 0; ks < ksteps; ks += 2) {
      stage_next(S1{}, ks);
      if constexpr (SYNC == 2) __syncthreads();
      if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
      mfma_phase(0);
      __syncthreads();
      if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
      stage_next(S0{}, ks + 1);
      if constexpr (SYNC == 2) __syncthreads();
      if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
      mfma_phase(1);
      __syncthreads();
      if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
    }
  }

  }

  // ---- epilogue: + bias, fp16, transpose through LDS, 16-byte coalesced NHWC stores
  constexpr int OS = BN + 8;
  half_t* Os = (half_t*)smem;                                // [BM][OS] (tiles dead: last barrier passed)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = wc * 64 + j * 16 + l15;
    const float bv = p.bias ? (float)p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) Os[(wr * IM * 16 + i * 16 + lg * 4 + r) * OS + col] = (half_t)(acc[i][j][r] + bv);
  }
  __syncthreads();
  for (int c = tid; c < BM * (BN / 8); c += NTH) {
    const int row = c / (BN / 8), ch = c % (BN / 8);
    int ho, wo;
    if (row_pos(row, ho, wo))
      *(half8*)(p.out + ((long)b * n_pos + ho * p.Wo + wo) * p.out_ld + ch * 8) = *(const half8*)(Os + row * OS + ch * 8);
  }
  // ---- GroupNorm / scale-attention statistics of this patch (what mq_dyconv_stats would re-read y from HBM for):
  // per channel sum, sum of squares and position-weighted sum of the fp16-rounded outputs; fixed summation order.
  if (p.stats) {
    constexpr int RPT = BM / (NTH / 32);                     // rows per thread (8 or 4)
    const int chunk = tid & 31, rg = tid >> 5;               // 8 channels x RPT rows per thread
    float s1[8], s2[8], s3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = s3[j] = 0.f;
    const float inv_n = 1.f / (float)n_pos;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int row = rg * RPT + k;
      int ho, wo;
      if (row_pos(row, ho, wo)) {
        const half8 v = *(const half8*)(Os + row * OS + chunk * 8);
        const float w = p.wy ? p.wy[ho] * p.wx[wo] : inv_n;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = (float)v[j];
          s1[j] += f; s2[j] += f * f; s3[j] += w * f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {                            // lanes l and l ^ 32 hold the same channels
      s1[j] += __shfl_xor(s1[j], 32); s2[j] += __shfl_xor(s2[j], 32); s3[j] += __shfl_xor(s3[j], 32);
    }
#if defined(MQ_F32)
    __syncthreads();                                         // fp32 O staging fills the LDS: the partials go over it once every row has been read
    float* red = (float*)smem;
#else
    float* red = (float*)(smem + (size_t)BM * OS * sizeof(half_t));      // [NW waves][32 chunks][24], behind the O staging
#endif
    if (lane < 32) {
      float* r = red + (wave * 32 + chunk) * 24;
#pragma unroll
      for (int j = 0; j < 8; ++j) { r[j] = s1[j]; r[8 + j] = s2[j]; r[16 + j] = s3[j]; }
    }
    __syncthreads();
    for (int idx = tid; idx < 32 * 24; idx += NTH) {
      const int lc = idx / 24, k = idx % 24;
      float a = 0.f;
#pragma unroll
      for (int g = 0; g < NW; ++g) a += red[(g * 32 + lc) * 24 + k];
      p.stats[(((long)b * tpi + trem) * BN + lc * 8 + (k & 7)) * 3 + (k >> 3)] = a;
    }
  }
}

// DCNv2 3x3, pad 1, 256 output channels.  x [B,H,W,C] fp16 NHWC (batch stride x_bs, C % 128 == 0), om [B,27,oH,oW] fp32
// (18 offsets + 9 mask logits -- or probabilities with flags bit 0 --, NCHW; flags bit 1: the caller promises all-zero offsets and mask 1, i.e. a plain conv), w [256, 9*C] fp16 (k = tap*C + c), bias [256] fp16 or NULL, out [B*Ho*Wo, out_ld];
// stats (optional) [B, mq_dcnv2_stats_blocks(H, W, stride), 256, 3] fp32 with position weights wy [Ho] x wx [Wo] (or NULL).
// tiles of 128 consecutive output positions in band order (default) or 8 x 16 patches (MQ_DCN_RASTER=0, the A/B switch): one process-wide choice, read once --
// the statistics buffers of the callers are sized by it
static bool dcn_raster_tiles() {
  static const bool r = [] { const char* e = getenv("MQ_DCN_RASTER"); return !(e && e[0] == '0'); }();
  return r;
}
#ifdef MQ_PRIMARY_UNIT
extern "C" int mq_dcnv2_stats_blocks(int H, int W, int stride) {
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  if (dcn_raster_tiles()) return (Ho * Wo + DCN_PH * DCN_PW - 1) / (DCN_PH * DCN_PW);
  return ((Ho + DCN_PH - 1) / DCN_PH) * ((Wo + DCN_PW - 1) / DCN_PW);
}
#endif

struct mq_dcn_branch {          // mirrors include/mqdet_hip.h
  const void* x; const float* om; const void* w; const void* bias; void* out; float* stats; const float* wy; const float* wx;
  long x_bs;
  int B, H, W, C, oH, oW, N, out_ld, stride, flags;
};

template <int NW, int SYNC, bool PLAIN, bool BDMA = false>
static int dcn_launch(dim3 grid, size_t smem, hipStream_t stream, const DcnGroup& g) {
  static MqOncePerDevice attr;
  if (attr.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)dcn_igemm8_kernel<NW, 0, SYNC, PLAIN, false, BDMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr.done();
  }
  hipLaunchKernelGGL((dcn_igemm8_kernel<NW, 0, SYNC, PLAIN, false, BDMA>), grid, dim3(64 * NW), smem, stream, g);
  return 0;
}

extern "C" int MQ_SYM(mq_dcnv2_group_fwd)(const mq_dcn_branch* br, int n, void* stream) {
  if (n <= 0) return 0;
  if (n > DCN_MAX_BRANCH) return -3;
  DcnGroup g;
  g.n = 0;
  g.first_tile[0] = 0;
  for (int i = 0; i < n; ++i) {
    const mq_dcn_branch& a = br[i];
    if (a.B <= 0) continue;
    if (a.N != 256 || a.C % 128 || a.stride < 1 || a.stride > 2 || a.out_ld < a.N || a.out_ld % 8) return -1;
    DcnFParams& p = g.br[g.n];
    p.x = (const half_t*)a.x; p.w = (const half_t*)a.w; p.bias = (const half_t*)a.bias; p.om = a.om; p.out = (half_t*)a.out;
    p.stats = a.stats; p.wy = a.wy; p.wx = a.wx;
    p.x_bs = a.x_bs; p.B = a.B; p.H = a.H; p.W = a.W; p.C = a.C; p.stride = a.stride; p.oH = a.oH; p.oW = a.oW; p.out_ld = a.out_ld; p.mask_prob = a.flags & 1;
    p.Ho = (a.H + 2 - 3) / a.stride + 1; p.Wo = (a.W + 2 - 3) / a.stride + 1;
    if ((long)p.Ho * p.Wo > (long)a.oH * a.oW) return -2;    // flat reads must stay inside the om buffer
    if (p.Ho > 32767 || p.Wo > 32767) return -1;             // (ho, wo) of a tile row are packed into one int
    if (dcn_raster_tiles()) {
      p.tiles_y = 0; p.tiles_x = (p.Ho * p.Wo + DCN_PH * DCN_PW - 1) / (DCN_PH * DCN_PW);      // tiles_y == 0: raster tiles, tiles_x per image
      p.tiles_total = a.B * p.tiles_x;
    } else {
      p.tiles_y = (p.Ho + DCN_PH - 1) / DCN_PH; p.tiles_x = (p.Wo + DCN_PW - 1) / DCN_PW;
      p.tiles_total = a.B * p.tiles_y * p.tiles_x;
    }
    g.first_tile[g.n + 1] = g.first_tile[g.n] + p.tiles_total;
    ++g.n;
  }
  if (g.n == 0) return 0;
  g.tiles_all = g.first_tile[g.n];
  constexpr size_t smem = dcn_smem_main() + 128 * sizeof(int);             // + the (ho, wo) table of the tile's rows
  static MqOncePerDevice attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)dcn_igemm8_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dcn_igemm8_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dcn_igemm8_kernel<16, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set.done();
  }
#if MQ_DCN_SPLIT
  // split-precise: 8 waves (wave tile 64 x 64: the planar hi / lo fragments are twice the registers of the fp16 ones -- 16 waves at 128 VGPRs spill)
  static const int nw = [] { const char* e = getenv("MQ_DCN_WAVES"); return (e && e[0] == '1') ? 16 : 8; }();
#else
  static const int nw = [] { const char* e = getenv("MQ_DCN_WAVES"); return (e && e[0] == '8') ? 8 : 16; }();   // A/B switch
#endif
  const dim3 grid((unsigned)(8 * ((g.tiles_all + 7) / 8)));
#ifdef MQ_PRIMARY_UNIT
  if (const int abl = (br[0].flags >> 8) & 15) {             // ablation timings (tools/microbench.py)
    switch (abl) {
#define MQ_DCN_ABL(A_)                                                                                                             \
      case A_: {                                                                                                                   \
        hipError_t e = hipFuncSetAttribute((const void*)dcn_igemm8_kernel<16, A_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        if (e != hipSuccess) return (int)e;                                                                                        \
        hipLaunchKernelGGL((dcn_igemm8_kernel<16, A_>), grid, dim3(1024), smem, (hipStream_t)stream, g);                            \
        break;                                                                                                                     \
      }
      MQ_DCN_ABL(1) MQ_DCN_ABL(2) MQ_DCN_ABL(3) MQ_DCN_ABL(4) MQ_DCN_ABL(8) MQ_DCN_ABL(7) MQ_DCN_ABL(15)
#undef MQ_DCN_ABL
      default: return -4;
    }
    MQ_CHECK_LAUNCH();
    return 0;
  }
#endif
  // one barrier per k-step by default since GPU calls 14 / 15 of round 3 (0.661 - 0.668 ms against 0.673 - 0.692 with two, three of three
  // comparisons; equal outputs); MQ_DCN_SYNC=2: the former schedule (A/B switch)
  static const int sync = [] { const char* e = getenv("MQ_DCN_SYNC"); return (e && e[0] == '2') ? 2 : 1; }();
  bool plain = true;                                          // flags bit 1 on EVERY branch: zero offsets, mask 1 (the caller's promise)
  for (int i = 0; i < n; ++i) plain = plain && (br[i].B <= 0 || (br[i].flags & 2));
  static const bool plain_on = [] { const char* e = getenv("MQ_DCN_PLAIN"); return !(e && e[0] == '0'); }();   // A/B switch
  {
    // flags bit 2 on EVERY branch: the weights are in LDS-tile order (ops.dcn_weight_tiles) -> the BDMA instantiations
    int nb = 0, nt = 0;
    for (int i = 0; i < n; ++i) if (br[i].B > 0) { ++nb; nt += (br[i].flags >> 2) & 1; }
    if (nt != 0 && nt != nb) return -5;
    if (nt) {
      constexpr int BW = MQ_DCN_SPLIT ? 8 : 16;               // waves of the shipped instantiation of this build
      int rc;
#if !MQ_DCN_SPLIT
      if (nw == 8)                                            // A/B switch MQ_DCN_WAVES=8: wave tiles of 64 x 64 (a third fewer fragment reads per k-step)
        rc = (plain && plain_on) ? dcn_launch<8, 1, true, true>(grid, smem, (hipStream_t)stream, g) : dcn_launch<8, 1, false, true>(grid, smem, (hipStream_t)stream, g);
      else
#endif
      rc = (plain && plain_on) ? dcn_launch<BW, 1, true, true>(grid, smem, (hipStream_t)stream, g)
                               : dcn_launch<BW, 1, false, true>(grid, smem, (hipStream_t)stream, g);
      if (rc) return rc;
      MQ_CHECK_LAUNCH();
      return 0;
    }
  }
#if MQ_DCN_SPLIT
  if (nw == 8) {                                              // split-precise default: 8 waves, one barrier per k-step, PLAIN for the FPN convs
    int rc;
    if (plain && plain_on) rc = dcn_launch<8, 1, true>(grid, smem, (hipStream_t)stream, g);
    else if (sync == 2) rc = dcn_launch<8, 2, false>(grid, smem, (hipStream_t)stream, g);
    else rc = dcn_launch<8, 1, false>(grid, smem, (hipStream_t)stream, g);
    if (rc) return rc;
    MQ_CHECK_LAUNCH();
    return 0;
  }
#endif
  if (plain && plain_on && nw == 16 && sync == 1) {
    static MqOncePerDevice attr_plain;
    if (attr_plain.first()) {
      hipError_t e = hipFuncSetAttribute((const void*)dcn_igemm8_kernel<16, 0, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      attr_plain.done();
    }
    hipLaunchKernelGGL((dcn_igemm8_kernel<16, 0, 1, true>), grid, dim3(1024), smem, (hipStream_t)stream, g);
  } else if (nw == 16 && sync == 1) hipLaunchKernelGGL((dcn_igemm8_kernel<16, 0, 1>), grid, dim3(1024), smem, (hipStream_t)stream, g);
  else if (nw == 16) hipLaunchKernelGGL(dcn_igemm8_kernel<16>, grid, dim3(1024), smem, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(dcn_igemm8_kernel<8>, grid, dim3(512), smem, (hipStream_t)stream, g);
  MQ_CHECK_LAUNCH();
  return 0;
}

extern "C" int MQ_SYM(mq_dcnv2_fwd)(const void* x, const float* om, const void* w, const void* bias, void* out, float* stats,
                            const float* wy, const float* wx, int B, int H, int W, int C, long x_bs, int oH, int oW, int N,
                            int out_ld, int stride, int flags, void* stream) {
  mq_dcn_branch a;
  a.x = x; a.om = om; a.w = w; a.bias = bias; a.out = out; a.stats = stats; a.wy = wy; a.wx = wx; a.x_bs = x_bs;
  a.B = B; a.H = H; a.W = W; a.C = C; a.oH = oH; a.oW = oW; a.N = N; a.out_ld = out_ld; a.stride = stride; a.flags = flags;
  return MQ_SYM(mq_dcnv2_group_fwd)(&a, 1, stream);
}

MQ_NAMESPACE_END
