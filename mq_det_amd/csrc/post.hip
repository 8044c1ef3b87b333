// Region-word alignment scoring and ATSS post-processing kernels for gfx950 (thin, HBM-bound).
//
// mq_align_scores_fwd -- reference rpn/vldyhead.py:884-887 (+bias, clamp +-50000) followed by
//   rpn/inference.py:656-683: sigmoid, token -> class aggregation (convert_grounding_to_od_logits[_v2], :772-824:
//   agg 0 = MEAN, 1 = MAX, 2 = POWER = prod^(1/n); ONEHOT is MEAN over the one-token index the host builds for it),
//   threshold 0.05, x sigmoid(centerness).  The reference builds a dense [B, HW, 3000] (LVIS) score
//   tensor of which <= 40 columns are non-zero; here only the L labels of the caption are produced.
//     dot    : [B, HW, T] fp16 = feat . (proj_tokens / exp(log_scale))^T       (library GEMM outside; batch stride
//              dot_bs elements: a level's slice of the all-level [B, N, T] product)
//     tbias  : [B, T] fp32     = emb . bias_lang + bias0
//     tokidx : [L, MT] int32   token positions of each label (-1 padded); batch stride tok_bs elements (0: one caption
//              shared by the batch; L*MT: one caption per batch item -- chunk batching of the LVIS protocol)
//     ctr    : [B, HW] fp16/fp32 centerness logits
//     out    : [B, HW, L] fp32 = (cls > thr) ? cls * sigmoid(ctr) : -1 ;  cls_out (optional) = cls
//   One wave per location: 256 token logits -> LDS, lanes then average their label's tokens.
//
// mq_box_decode -- BoxCoder.decode (vldyhead.py:78-108), clip_to_image (bounding_box.py:221-232), sqrt
//   score (inference.py:707), label = class_idx + 1 (:696) for the top-k candidates of one level.
//
// mq_ml_nms -- class-aware NMS, reference csrc/cuda/ml_nms.cu: IoU with the legacy +1 widths, 0 across
//   labels (:15-26), 64x64 bitmask tiles (:28-75).  The reference copies the mask to the host and sweeps
//   serially there (:117-140, a device sync per image); here the sweep runs on the device, one wave per
//   image, 64 boxes at a time (intra-chunk dependencies via wave shuffles).
#include "common.h"

MQ_NAMESPACE_BEGIN

template <typename TD>
__global__ __launch_bounds__(256) void align_scores_kernel(const TD* __restrict__ dot, const float* __restrict__ tbias,
                                                           const int* __restrict__ tokidx, const half_t* __restrict__ ctr,
                                                           float* __restrict__ out, float* __restrict__ cls_out,
                                                           int B, int HW, int T, int L, int MT, float thr, long dot_bs, long tok_bs,
                                                           int agg) {
  extern __shared__ float sig[];                 // [4][T]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long loc = (long)blockIdx.x * 4 + wave;
  if (loc >= (long)B * HW) return;
  const int b = loc / HW;
  const TD* drow = dot + (long)b * dot_bs + (loc - (long)b * HW) * T;
  const int* tix = tokidx + (long)b * tok_bs;
  float* sw = sig + wave * T;
  if ((T & 3) == 0 && (dot_bs & 3) == 0) {       // 8-byte (fp16) / 16-byte (fp32) loads: one per lane for T = 256
    for (int t = lane * 4; t < T; t += 256) {
      float d[4];
      if constexpr (sizeof(TD) == 2) {
        const half4 h = *(const half4*)(drow + t);
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = (float)h[j];
      } else {
        const float4_ f = *(const float4_*)(drow + t);
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = f[j];
      }
      const float4_ tb = *(const float4_*)(tbias + (long)b * T + t);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = d[j] + tb[j];
        v = fminf(fmaxf(v, -50000.f), 50000.f);
        sw[t + j] = 1.f / (1.f + __expf(-v));
      }
    }
  } else {
    for (int t = lane; t < T; t += 64) {
      float v = (float)drow[t] + tbias[(long)b * T + t];
      v = fminf(fmaxf(v, -50000.f), 50000.f);
      sw[t] = 1.f / (1.f + __expf(-v));
    }
  }
  wave_lds_fence();
  const float c = 1.f / (1.f + __expf(-(float)ctr[loc]));
  for (int l = lane; l < L; l += 64) {
    float s = agg == 2 ? 1.f : 0.f;
    int n = 0;
    for (int j = 0; j < MT; ++j) {
      int t = tix[l * MT + j];
      if (t >= 0) {
        const float v = sw[t];
        s = agg == 0 ? s + v : (agg == 1 ? fmaxf(s, v) : s * v);          // sigmoid outputs are >= 0: 0 is the identity of max
        ++n;
      }
    }
    float cls = 0.f;                                    // a label without tokens scores 0 (never a candidate)
    if (n > 0) cls = agg == 0 ? s / (float)n : (agg == 1 ? s : powf(s, 1.f / (float)n));
    if (cls_out) cls_out[loc * L + l] = cls;
    // candidates are decided by the class score alone (rpn/inference.py:677): keep them > 0 even if the product with a
    // vanishing centerness underflows, so that "value > 0" identifies a candidate downstream
    out[loc * L + l] = cls > thr ? fmaxf(cls * c, 1.17549435e-38f) : -1.f;
  }
}

extern "C" int MQ_SYM(mq_align_scores_fwd)(const void* dot, int dot_f32, const float* tbias, const int* tokidx, long tok_bs,
                                   const void* ctr, float* out, float* cls_out, int B, int HW, int T, int L, int MT, float thr,
                                   long dot_bs, int agg, void* stream) {
  if (B <= 0 || HW <= 0 || L <= 0) return 0;
  if (agg < 0 || agg > 2) return -1;
  long locs = (long)B * HW;
  const dim3 grid((unsigned)((locs + 3) / 4));
  const long dbs = dot_bs > 0 ? dot_bs : (long)HW * T;
  if (dot_f32)
    hipLaunchKernelGGL(align_scores_kernel<float>, grid, dim3(256), 4 * T * sizeof(float), (hipStream_t)stream, (const float*)dot,
                       tbias, tokidx, (const half_t*)ctr, out, cls_out, B, HW, T, L, MT, thr, dbs, tok_bs, agg);
  else
    hipLaunchKernelGGL(align_scores_kernel<half_t>, grid, dim3(256), 4 * T * sizeof(float), (hipStream_t)stream, (const half_t*)dot,
                       tbias, tokidx, (const half_t*)ctr, out, cls_out, B, HW, T, L, MT, thr, dbs, tok_bs, agg);
  MQ_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
template <typename TR>
__global__ void box_decode_kernel(const float* __restrict__ val, const long* __restrict__ flat, const TR* __restrict__ reg,
                                  const float* __restrict__ anchors, const int* __restrict__ label_ids,
                                  const float* __restrict__ im_wh, float* __restrict__ boxes, float* __restrict__ scores,
                                  int* __restrict__ labels, int B, int K, int HW, int L, long out_stride, long out_off, long lab_bs) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * K) return;
  const int b = i / K, k = i % K;
  const long o = (long)b * out_stride + out_off + k;
  const float v = val[i];
  if (!(v > 0.f)) {                       // not a candidate (score -1 filler)
    scores[o] = -1.f; labels[o] = 0;
    boxes[o * 4 + 0] = boxes[o * 4 + 1] = boxes[o * 4 + 2] = boxes[o * 4 + 3] = 0.f;
    return;
  }
  const long f = flat[i];
  const int loc = f / L, l = f % L;
  const TR* r = reg + ((long)b * HW + loc) * 4;
  const float* a = anchors + (long)loc * 4;
  const float w = a[2] - a[0] + 1.f, h = a[3] - a[1] + 1.f;
  const float cx = (a[2] + a[0]) * 0.5f, cy = (a[3] + a[1]) * 0.5f;
  const float lim = 4.135166556742356f;   // log(1000/16)
  const float dx = (float)r[0] / 10.f, dy = (float)r[1] / 10.f;
  const float dw = fminf((float)r[2] / 5.f, lim), dh = fminf((float)r[3] / 5.f, lim);
  const float pcx = dx * w + cx, pcy = dy * h + cy;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  const float W = im_wh[b * 2 + 0], H = im_wh[b * 2 + 1];
  boxes[o * 4 + 0] = fminf(fmaxf(pcx - 0.5f * (pw - 1.f), 0.f), W - 1.f);
  boxes[o * 4 + 1] = fminf(fmaxf(pcy - 0.5f * (ph - 1.f), 0.f), H - 1.f);
  boxes[o * 4 + 2] = fminf(fmaxf(pcx + 0.5f * (pw - 1.f), 0.f), W - 1.f);
  boxes[o * 4 + 3] = fminf(fmaxf(pcy + 0.5f * (ph - 1.f), 0.f), H - 1.f);
  scores[o] = sqrtf(v);
  labels[o] = label_ids[(long)b * lab_bs + l];
}

extern "C" int MQ_SYM(mq_box_decode)(const float* val, const long* flat, const void* reg, int reg_f32, const float* anchors, const int* label_ids,
                             long lab_bs, const float* im_wh, float* boxes, float* scores, int* labels, int B, int K, int HW,
                             int L, long out_stride, long out_off, void* stream) {
  if (B <= 0 || K <= 0) return 0;
  long n = (long)B * K;
  if (reg_f32)
    hipLaunchKernelGGL(box_decode_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, val, flat,
                       (const float*)reg, anchors, label_ids, im_wh, boxes, scores, labels, B, K, HW, L, out_stride, out_off, lab_bs);
  else
    hipLaunchKernelGGL(box_decode_kernel<half_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, val, flat,
                       (const half_t*)reg, anchors, label_ids, im_wh, boxes, scores, labels, B, K, HW, L, out_stride, out_off, lab_bs);
  MQ_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
#ifdef MQ_PRIMARY_UNIT                                     // NMS works on fp32 boxes: one copy, in the fp16 translation unit
__device__ __forceinline__ float ml_iou(const float* a, int la, const float* b, int lb) {
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
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ labels,
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
      if (ml_iou(a, la, cb + j * 4, cl[j]) > thr) t |= 1ULL << j;
  }
  if (i < N) mask[((long)b * N + i) * col_blocks + col_blk] = t;
}

// one wave per image; lane owns remv words {lane, lane+64, ...}
template <int SLOTS>
__global__ __launch_bounds__(64) void nms_sweep_kernel(const unsigned long long* __restrict__ mask,
                                                       const int* __restrict__ nvalid, unsigned char* __restrict__ keep,
                                                       int N, int col_blocks) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = nvalid[b];
  const unsigned long long* mb = mask + (long)b * N * col_blocks;
  unsigned char* kb = keep + (long)b * N;
  unsigned long long remv[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) remv[s] = 0ULL;
  const int nchunks = (n + 63) / 64;
  for (int c = 0; c < nchunks; ++c) {
    const int i = c * 64 + lane;
    // word c of remv lives in lane (c % 64), slot (c / 64)
    unsigned long long word = 0ULL;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
      if (s == c / 64) word = remv[s];
    word = __shfl(word, c % 64);
    unsigned long long diag = (i < n) ? mb[(long)i * col_blocks + c] : 0ULL;
    int alive = (i < n) && !((word >> lane) & 1ULL);
    for (int j = 0; j < 64; ++j) {
      int aj = __shfl(alive, j);
      unsigned long long dj = __shfl(diag, j);
      if (aj && ((dj >> lane) & 1ULL)) alive = 0;          // only bits > j are ever set in row j's diagonal word
    }
    if (i < N) kb[i] = (unsigned char)alive;
    unsigned long long alive_mask = __ballot(alive);
    while (alive_mask) {
      int r = __ffsll((long long)alive_mask) - 1;
      alive_mask &= alive_mask - 1;
      const unsigned long long* row = mb + (long)(c * 64 + r) * col_blocks;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        int w = lane + 64 * s;
        if (w > c && w < col_blocks) remv[s] |= row[w];
      }
    }
  }
  for (int i = nchunks * 64 + lane; i < N; i += 64) kb[i] = 0;
}

// Same sweep, five waves per image: waves 1-4 stream the 64-row block of mask words of chunk c + 2 into LDS (coalesced
// 16-byte loads, all of a wave's loads in flight before the first store, triple-buffered) while wave 0 resolves chunk c
// out of LDS.  The one-wave version above follows every surviving row with a dependent global load (~0.5 us each,
// ~5000 candidates): 1.0 ms per forward, all of it on the critical path.
template <int SLOTS>
__global__ __launch_bounds__(320) void nms_sweep_lds_kernel(const unsigned long long* __restrict__ mask,
                                                            const int* __restrict__ nvalid, unsigned char* __restrict__ keep,
                                                            int N, int col_blocks) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long rows_s[];      // [3][64 * col_blocks]
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
  __syncthreads();
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
  }
  if (wave == 0)
    for (int i = nchunks * 64 + lane; i < N; i += 64) kb[i] = 0;
}

template <int SLOTS>
static int launch_sweep_lds(const unsigned long long* mask, const int* nvalid, unsigned char* keep, int B, int N, int col_blocks,
                            hipStream_t stream) {
  const size_t smem = (size_t)3 * 64 * col_blocks * sizeof(unsigned long long);
  static MqMaxPerDevice attr_set;
  if (attr_set.need(smem)) {
    hipError_t e = hipFuncSetAttribute((const void*)nms_sweep_lds_kernel<SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set.done(smem);
  }
  hipLaunchKernelGGL((nms_sweep_lds_kernel<SLOTS>), dim3(B), dim3(320), smem, stream, mask, nvalid, keep, N, col_blocks);
  MQ_CHECK_LAUNCH();
  return 0;
}

extern "C" long mq_ml_nms_workspace_bytes(int B, int N) {
  long col_blocks = (N + 63) / 64;
  return (long)B * N * col_blocks * 8;
}

extern "C" int mq_ml_nms(const float* boxes, const int* labels, const int* nvalid, void* workspace, unsigned char* keep,
                         int B, int N, float thr, void* stream) {
  if (B <= 0 || N <= 0) return 0;
  int col_blocks = (N + 63) / 64;
  if (col_blocks > 256) return -1;
  unsigned long long* mask = (unsigned long long*)workspace;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(col_blocks, col_blocks, B), dim3(64), 0, (hipStream_t)stream, boxes, labels,
                     nvalid, mask, N, col_blocks, thr);
  MQ_CHECK_LAUNCH();
  if (col_blocks <= 64)                                    // three 64-row blocks of mask words fit the 160 KB LDS
    return launch_sweep_lds<1>(mask, nvalid, keep, B, N, col_blocks, (hipStream_t)stream);
  else if (col_blocks <= 104)
    return launch_sweep_lds<2>(mask, nvalid, keep, B, N, col_blocks, (hipStream_t)stream);
  else if (col_blocks <= 128)
    hipLaunchKernelGGL((nms_sweep_kernel<2>), dim3(B), dim3(64), 0, (hipStream_t)stream, mask, nvalid, keep, N, col_blocks);
  else
    hipLaunchKernelGGL((nms_sweep_kernel<4>), dim3(B), dim3(64), 0, (hipStream_t)stream, mask, nvalid, keep, N, col_blocks);
  MQ_CHECK_LAUNCH();
  return 0;
}
#endif

MQ_NAMESPACE_END
