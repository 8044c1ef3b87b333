// mq_ml_nms_topk: class-aware NMS (post.hip: mq_ml_nms) that STOPS once `max_keep` boxes of an image are kept.
//
// The boxes arrive sorted by score and the caller only uses the max_keep highest-scoring survivors (ATSSPostProcessor.select_over_all_levels,
// rpn/inference.py:748-769: NMS, then kthvalue / top DETECTIONS_PER_IMG): the first max_keep survivors of the sweep are exactly those, every
// later box is irrelevant.  mq_ml_nms resolves all ~5000 candidates of an image, 64 at a time in one wave -- a serial chain of ~80
// chunks, 0.3 ms at the very end of the step where nothing overlaps it (nms_sweep_lds_kernel, profiles/r02_call5); with 300 detections
// per image the sweep typically ends after 10 - 20 chunks.  keep[] is 0 for every box after the stopping chunk; the selected
// detections are identical to mq_ml_nms + top-k.  Same mask kernel and sweep as post.hip (copied: post.o stays the object that was
// validated on the device) plus a stop flag in LDS read by all five waves after the per-chunk barrier.
// Default since round 3 (KERNELS["NMS_EARLY_STOP"] = 1; device parity: tests/test_gpu_parity.py test_opt_in_kernel[nms_early_stop]).
#include "common.h"

MQ_NAMESPACE_BEGIN
#ifdef MQ_PRIMARY_UNIT                                     // NMS works on fp32 boxes: one copy, in the fp16 translation unit

__device__ __forceinline__ float ml_iou2(const float* a, int la, const float* b, int lb) {
  if (la != lb) return 0.f;
  float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  float w = fmaxf(right - left + 1.f, 0.f), h = fmaxf(bottom - top + 1.f, 0.f);
  float inter = w * h;
  float sa = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
  float sb = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
  return inter / (sa + sb - inter);
}

// boxes sorted by score (descending) per image; rows >= nvalid[b] are ignored.
__global__ __launch_bounds__(64) void nms2_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ labels,
                                                      const int* __restrict__ nvalid, unsigned long long* __restrict__ mask,
                                                      int N, int col_blocks, float thr) {
  const int b = blockIdx.z, row_blk = blockIdx.y, col_blk = blockIdx.x;
  const int n = nvalid[b];
  if (row_blk * 64 >= n || col_blk * 64 >= n || col_blk < row_blk) {
    // upper-triangular only; untouched words must still read as 0
    int i = row_blk * 64 + threadIdx.x;
    if (i < N) mask[((long)b * N + i) * col_blocks + col_blk] = 0ULL;
    return;
  }
  __shared__ float cb[64 * 4];
  __shared__ int cl[64];
  const float* bb = boxes + (long)b * N * 4;
  const int* lb = labels + (long)b * N;
  const int cj = col_blk * 64 + threadIdx.x;
  if (cj < n) {
    cb[threadIdx.x * 4 + 0] = bb[cj * 4 + 0]; cb[threadIdx.x * 4 + 1] = bb[cj * 4 + 1];
    cb[threadIdx.x * 4 + 2] = bb[cj * 4 + 2]; cb[threadIdx.x * 4 + 3] = bb[cj * 4 + 3];
    cl[threadIdx.x] = lb[cj];
  }
  __syncthreads();
  const int i = row_blk * 64 + threadIdx.x;
  unsigned long long t = 0ULL;
  if (i < n) {
    float a[4] = {bb[i * 4 + 0], bb[i * 4 + 1], bb[i * 4 + 2], bb[i * 4 + 3]};
    int la = lb[i];
    int cols = min(64, n - col_blk * 64);
    int start = (row_blk == col_blk) ? threadIdx.x + 1 : 0;
    for (int j = start; j < cols; ++j)
      if (ml_iou2(a, la, cb + j * 4, cl[j]) > thr) t |= 1ULL << j;
  }
  if (i < N) mask[((long)b * N + i) * col_blocks + col_blk] = t;
}

// Same sweep, five waves per image: waves 1-4 stream the 64-row block of mask words of chunk c + 2 into LDS (coalesced
// 16-byte loads, all of a wave's loads in flight before the first store, triple-buffered) while wave 0 resolves chunk c
// out of LDS.  The one-wave version above follows every surviving row with a dependent global load (~0.5 us each,
// ~5000 candidates): 1.0 ms per forward, all of it on the critical path.
template <int SLOTS>
__global__ __launch_bounds__(320) void nms2_sweep_kernel(const unsigned long long* __restrict__ mask,
                                                            const int* __restrict__ nvalid, unsigned char* __restrict__ keep,
                                                            int N, int col_blocks, int max_keep) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long rows_s[];      // [3][64 * col_blocks]
  __shared__ int stop_s[2];     // [c & 1]: set by wave 0 in iteration c once max_keep boxes are kept.  Two slots: wave 0 may be an
                                // iteration ahead of a loader wave that has not yet read the flag of the previous one
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = nvalid[b];
  const unsigned long long* mb = mask + (long)b * N * col_blocks;
  unsigned char* kb = keep + (long)b * N;
  const int nchunks = (n + 63) / 64;
  const int blk = 64 * col_blocks;                          // words per chunk block (even)
  auto stage = [&](int c) {                                 // loader waves: chunk c -> buffer c % 3
    if (c >= nchunks) return;
    const unsigned long long* src = mb + (long)c * blk;
    const int avail = (min(N, c * 64 + 64) - c * 64) * col_blocks;       // words of rows that exist in the workspace
    unsigned long long* dst = rows_s + (c % 3) * blk;
    const int lt = (wave - 1) * 64 + lane;                  // 0..255
    constexpr int U = 8;
    for (int base = 0; base < blk; base += 256 * 2 * U) {
      u64x2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + (u * 256 + lt) * 2;
        v[u] = (u64x2){0ULL, 0ULL};
        if (idx + 1 < avail) v[u] = *(const u64x2*)(src + idx);
        else if (idx < avail) v[u][0] = src[idx];             // odd tail: never read past the workspace
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + (u * 256 + lt) * 2;
        if (idx < blk) *(u64x2*)(dst + idx) = v[u];
      }
    }
  };
  if (wave > 0) { stage(0); stage(1); }
  if (threadIdx.x < 2) stop_s[threadIdx.x] = 0;
  __syncthreads();
  int kept = 0, c_end = nchunks;                            // c_end: first chunk that was not resolved
  unsigned long long remv[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) remv[s] = 0ULL;
  for (int c = 0; c < nchunks; ++c) {
    if (wave > 0) {
      stage(c + 2);                                         // buffer (c + 2) % 3 was last read for chunk c - 1
    } else {
      const unsigned long long* L = rows_s + (c % 3) * blk;
      const int i = c * 64 + lane;
      unsigned long long word = 0ULL;                      // word c of remv lives in lane (c % 64), slot (c / 64)
#pragma unroll
      for (int s = 0; s < SLOTS; ++s)
        if (s == c / 64) word = remv[s];
      word = __shfl(word, c % 64);
      const unsigned long long diag = (i < n) ? L[lane * col_blocks + c] : 0ULL;
      int alive = (i < n) && !((word >> lane) & 1ULL);
      for (int j = 0; j < 64; ++j) {
        const int aj = __shfl(alive, j);
        const unsigned long long dj = __shfl(diag, j);
        if (aj && ((dj >> lane) & 1ULL)) alive = 0;        // only bits > j are ever set in row j's diagonal word
      }
      if (i < N) kb[i] = (unsigned char)alive;
      const unsigned long long alive_mask = __ballot(alive);
      // boxes are sorted by score: the first max_keep survivors ARE the top max_keep of the kept set -- nothing that comes later can
      // enter the final selection (rpn/inference.py:757-766 keeps the max_keep highest scores of the NMS output), so the sweep ends here
      kept += __popcll(alive_mask);
      if (kept >= max_keep && lane == 0) stop_s[c & 1] = 1;
      // OR the rows of the survivors into remv: LDS reads are issued 8 rows at a time, the test is wave-uniform.
      // (words <= c are OR-ed too: they belong to chunks already resolved and are never read again)
      for (int r0 = 0; r0 < 64; r0 += 8) {
        if (((alive_mask >> r0) & 0xFFULL) == 0ULL) continue;
        unsigned long long v[8][SLOTS];
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int s = 0; s < SLOTS; ++s) {
            const int w = lane + 64 * s;
            v[k][s] = (w < col_blocks) ? L[(r0 + k) * col_blocks + w] : 0ULL;
          }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if ((alive_mask >> (r0 + k)) & 1ULL) {
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) remv[s] |= v[k][s];
          }
      }
    }
    __syncthreads();
    if (stop_s[c & 1]) { c_end = c + 1; break; }                   // block-uniform: every wave reads the flag after the same barrier
  }
  if (wave == 0)
    for (int i = c_end * 64 + lane; i < N; i += 64) kb[i] = 0;
}

template <int SLOTS>
static int launch_sweep2(const unsigned long long* mask, const int* nvalid, unsigned char* keep, int B, int N, int col_blocks, int max_keep,
                         hipStream_t stream) {
  const size_t smem = (size_t)3 * 64 * col_blocks * sizeof(unsigned long long);
  static MqMaxPerDevice attr_set;
  if (attr_set.need(smem)) {
    hipError_t e = hipFuncSetAttribute((const void*)nms2_sweep_kernel<SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set.done(smem);
  }
  hipLaunchKernelGGL((nms2_sweep_kernel<SLOTS>), dim3(B), dim3(320), smem, stream, mask, nvalid, keep, N, col_blocks, max_keep);
  MQ_CHECK_LAUNCH();
  return 0;
}

// boxes [B,N,4] fp32 sorted by score (descending) per image, labels [B,N] int32, nvalid [B] int32, workspace of mq_ml_nms_workspace_bytes(B, N)
// bytes -> keep [B,N] uint8; max_keep >= 1.  N <= 104 * 64 = 6656 (the LDS sweep); larger inputs: -1 (use mq_ml_nms).
extern "C" int mq_ml_nms_topk(const float* boxes, const int* labels, const int* nvalid, void* workspace, unsigned char* keep,
                              int B, int N, float thr, int max_keep, void* stream) {
  if (B <= 0 || N <= 0) return 0;
  if (max_keep < 1) return -2;
  const int col_blocks = (N + 63) / 64;
  if (col_blocks > 104) return -1;
  unsigned long long* mask = (unsigned long long*)workspace;
  hipLaunchKernelGGL(nms2_mask_kernel, dim3(col_blocks, col_blocks, B), dim3(64), 0, (hipStream_t)stream, boxes, labels, nvalid, mask, N,
                     col_blocks, thr);
  MQ_CHECK_LAUNCH();
  if (col_blocks <= 64) return launch_sweep2<1>(mask, nvalid, keep, B, N, col_blocks, max_keep, (hipStream_t)stream);
  return launch_sweep2<2>(mask, nvalid, keep, B, N, col_blocks, max_keep, (hipStream_t)stream);
}
#endif

MQ_NAMESPACE_END
