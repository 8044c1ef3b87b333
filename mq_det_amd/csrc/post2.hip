// ATSS post-processing in three launches (round 4) -- ATSSPostProcessor.forward_for_single_feature_map / select_over_all_levels
// (reference rpn/inference.py:620-769) between the alignment kernel and the NMS, gfx950:
//
//   mq_post_select_fwd    per (image, level): the PRE_NMS_TOP_N best candidates of the level's score map (an exact radix select over the
//                         fp32 scores, ties at the cut resolved by the smaller flat index), decoded right away (BoxCoder.decode,
//                         clip_to_image, sqrt score, label id -- vldyhead.py:78-108, inference.py:696-707) into the image's candidate list
//   mq_post_sort_fwd      per image: the candidate list sorted by (score descending, candidate id ascending) -- the order ml_nms sweeps in
//   mq_post_finalize_fwd  per image, after mq_ml_nms_topk: the first DETECTIONS_PER_IMG kept detections + those tied with the last of
//                         them (inference.py:757-766: kthvalue + `>=`) as one packed [K2, 6] block, the live count and the overflow flag
//
// Before: 5 x torch.topk(1000) (multi-block radix select: 13 launches each on the large levels), 5 x box_decode, argsort, 3 gathers,
// a final topk and the tie logic as ~145 launches of 3 .. 80 us in ONE dependent chain at the very end of the forward, where nothing
// overlaps them: 1.1 ms of kernel time, 1.7 ms on the timeline (profiles/r04_call2_timeline_tail.txt).  All integer / index work: the
// selection is exact, the outputs are bit-identical to the chain they replace up to the order of EXACTLY equal scores (the reference's
// topk / sort are not stable there either); here that order is fixed by the candidate id, so results are reproducible run to run.
#include "common.h"

#ifdef MQ_PRIMARY_UNIT      // fp32 / integer data only: one copy, in the fp16 translation unit

namespace {
constexpr int PS_MAXLVL = 8, PS_NT = 1024, PS_NW = PS_NT / 64;
constexpr int SORT_MAX = 8192;

constexpr int PS_MAXSEG = 64, PS_SEGLEN = 32768, PS_KMAX = 2048;

// One radix-select problem per (segment, image).  Launch 1 ("local"): a segment = a slice of <= PS_SEGLEN scores of one level; its k best
// (value, flat index) pairs go to the scratch lists.  Launch 2 ("merge"): a segment = a level; its input is the concatenation of the
// level's slice lists (the k best of the level are among the k best of every slice), its output the level's k best, SORTED and decoded.
// Why two launches: one workgroup scanning a P3 score map (B = 8: 672 000 values per image, 4-5 passes) took 186 us -- a workgroup has
// ~64 KB of loads in flight; 21 slices of it run on 21 CUs at once.
struct PostSelectParams {
  const float* ranked[PS_MAXLVL];   // level l: [B, hw[l], L] fp32; a candidate has value > 0
  const float* reg[PS_MAXLVL];      // level l: [B, hw[l], 4] fp32 box deltas
  const float* anchors[PS_MAXLVL];  // level l: [hw[l], 4]
  int hw[PS_MAXLVL], k[PS_MAXLVL], off[PS_MAXLVL];   // locations, candidates kept, first slot of the level in the image's list
  int idbase[PS_MAXLVL];            // first candidate id of the level (ids = idbase + flat index: unique per image)
  int coff[PS_MAXLVL], cn[PS_MAXLVL];                // the level's slice lists in the scratch: first entry, number of entries
  short seg_lvl[PS_MAXSEG];         // local segments: level,
  int seg_start[PS_MAXSEG], seg_len[PS_MAXSEG], seg_k[PS_MAXSEG], seg_coff[PS_MAXSEG];   // first flat index, length, kept, scratch offset
  float* cval; int* cidx;           // scratch [B, ctot]: value (-1 = empty) and flat index of the slice winners
  const int* label_ids; long lab_bs;
  const float* im_wh;               // [B, 2] (w, h)
  float* boxes; float* scores; int* labels; int* ids;     // [B, tot, 4], [B, tot], [B, tot], [B, tot]
  int B, L, NL, tot, nseg, ctot;
};

// block-wide helpers (1024 threads)
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl(v, lane - o);
    if (lane >= o) v += t;
  }
  return v;
}

// Exclusive prefix of `v` over the threads of the block in thread order; *total = block sum.  `ws` = PS_NW + 1 words of LDS.
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned* ws, unsigned* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned inc = wave_incl_scan(v, lane);
  __syncthreads();                                     // ws may still be read from the previous call
  if (lane == 63) ws[wave] = inc;
  __syncthreads();
  if (threadIdx.x < 64) {
    const unsigned w = lane < PS_NW ? ws[lane] : 0u;
    const unsigned wi = wave_incl_scan(w, lane);
    if (lane < PS_NW) ws[lane] = wi - w;
    if (lane == PS_NW - 1) ws[PS_NW] = wi;
  }
  __syncthreads();
  *total = ws[PS_NW];
  return ws[wave] + inc - v;
}

// One digit of a radix select.  hist[NB] holds the digit counts of the elements still in play; find the digit d* that holds the
// `want`-th largest element (counting from the largest digit down) -> out[0] = d*, out[1] = number of elements with a LARGER digit,
// out[2] = hist[d*].  Caller guarantees sum(hist) >= want >= 1.  NB <= 2 * PS_NT.
template <int NB>
__device__ __forceinline__ void pick_digit(const unsigned* hist, unsigned want, unsigned* ws, unsigned* out) {
  // thread t owns the reversed positions 2t, 2t + 1 (reversed position r <-> digit NB - 1 - r)
  const int t = threadIdx.x;
  unsigned h0 = 0, h1 = 0;
  if (2 * t < NB) h0 = hist[NB - 1 - 2 * t];
  if (2 * t + 1 < NB) h1 = hist[NB - 2 - 2 * t];
  unsigned total;
  const unsigned before = block_excl_scan(h0 + h1, ws, &total);
  if (before < want && before + h0 >= want && h0) { out[0] = NB - 1 - 2 * t; out[1] = before; out[2] = h0; }
  else if (before + h0 < want && before + h0 + h1 >= want && h1) { out[0] = NB - 2 - 2 * t; out[1] = before + h0; out[2] = h1; }
  __syncthreads();
}

// for_each_element(f): f(index, value) for the n floats at src, 16-byte loads where the row allows, 4 loads in flight per thread
template <class F>
__device__ __forceinline__ void scan_values(const float* __restrict__ src, int n, F f) {
  if ((((uintptr_t)src) & 15) == 0) {
    const int n4 = n >> 2;
    const float4_* s4 = (const float4_*)src;
    int i = threadIdx.x;
    for (; i + 3 * PS_NT < n4; i += 4 * PS_NT) {
      const float4_ a = s4[i], b = s4[i + PS_NT], c = s4[i + 2 * PS_NT], d = s4[i + 3 * PS_NT];
#pragma unroll
      for (int j = 0; j < 4; ++j) { f(4 * i + j, a[j]); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { f(4 * (i + PS_NT) + j, b[j]); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { f(4 * (i + 2 * PS_NT) + j, c[j]); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { f(4 * (i + 3 * PS_NT) + j, d[j]); }
    }
    for (; i < n4; i += PS_NT) {
      const float4_ a = s4[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) { f(4 * i + j, a[j]); }
    }
    for (int r = 4 * n4 + threadIdx.x; r < n; r += PS_NT) f(r, src[r]);
  } else {
    for (int r = threadIdx.x; r < n; r += PS_NT) f(r, src[r]);
  }
}

__device__ __forceinline__ unsigned key_of(float v) { return v > 0.f ? __float_as_uint(v) : 0u; }   // candidates: positive floats, monotone as uint
}  // namespace

// Radix select of the K largest keys of one segment (11 + 10 + 10 bits of the fp32 pattern), ties at the cut by the smaller flat index
// (11 + 11 bits, only when the cut falls inside a group of equal keys).  MERGE = false: values = a slice of a level's score map, flat index
// = position; the winners go to the scratch lists in arbitrary order.  MERGE = true: values / flat indices = the level's slice lists; the
// winners are sorted by (value descending, flat index ascending) in LDS, decoded and written to the level's slots in that order.
template <bool MERGE>
__global__ __launch_bounds__(PS_NT) void post_select_kernel(PostSelectParams p) {
  __shared__ unsigned hist[2048];
  __shared__ unsigned ws[PS_NW + 1];
  __shared__ unsigned pick[3];
  __shared__ unsigned counter;
  __shared__ unsigned long long sel[MERGE ? PS_KMAX : 1];
  const int seg = blockIdx.x, b = blockIdx.y;
  const int lvl = MERGE ? seg : p.seg_lvl[seg];
  const int n = MERGE ? p.cn[lvl] : p.seg_len[seg];
  const int K = MERGE ? p.k[lvl] : p.seg_k[seg];
  const float* src = MERGE ? p.cval + (long)b * p.ctot + p.coff[lvl] : p.ranked[lvl] + (long)b * p.hw[lvl] * p.L + p.seg_start[seg];
  const int* sidx = MERGE ? p.cidx + (long)b * p.ctot + p.coff[lvl] : nullptr;
  const int start = MERGE ? 0 : p.seg_start[seg];
  const int tid = threadIdx.x;
  auto flat_of = [&](int i) -> unsigned { return MERGE ? (unsigned)sidx[i] : (unsigned)(start + i); };
  auto zero_hist = [&](int nb) {
    for (int i = tid; i < nb; i += PS_NT) hist[i] = 0u;
    __syncthreads();
  };
  // ---- digit 0: key bits 30..20
  zero_hist(2048);
  scan_values(src, n, [&](int, float v) { const unsigned k = key_of(v); if (k) atomicAdd(&hist[k >> 20], 1u); });
  __syncthreads();
  unsigned total;
  {
    unsigned h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
    (void)block_excl_scan(h0 + h1, ws, &total);
  }
  unsigned kth = 0u, need_eq = 0u, cnt_eq = 0u;          // select keys > kth, and need_eq of the cnt_eq keys == kth
  if (total > (unsigned)K) {
    pick_digit<2048>(hist, (unsigned)K, ws, pick);
    const unsigned d0 = pick[0];
    unsigned want = (unsigned)K - pick[1];
    __syncthreads();
    // ---- digit 1: bits 19..10 of the keys whose digit 0 is d0
    zero_hist(1024);
    scan_values(src, n, [&](int, float v) { const unsigned k = key_of(v); if (k && (k >> 20) == d0) atomicAdd(&hist[(k >> 10) & 1023u], 1u); });
    __syncthreads();
    pick_digit<1024>(hist, want, ws, pick);
    const unsigned d1 = pick[0];
    want -= pick[1];
    __syncthreads();
    // ---- digit 2: bits 9..0
    const unsigned pre = (d0 << 10) | d1;
    zero_hist(1024);
    scan_values(src, n, [&](int, float v) { const unsigned k = key_of(v); if (k && (k >> 10) == pre) atomicAdd(&hist[k & 1023u], 1u); });
    __syncthreads();
    pick_digit<1024>(hist, want, ws, pick);
    kth = (pre << 10) | pick[0];
    need_eq = want - pick[1];
    cnt_eq = pick[2];
    __syncthreads();
  }
  // ---- ties at the cut: the need_eq SMALLEST flat indices among the cnt_eq keys equal to kth
  unsigned idx_thr = 0xFFFFFFFFu;                        // ties with flat index <= idx_thr are taken
  if (need_eq < cnt_eq) {
    // need_eq >= 1 here (the cut digit holds the want-th element).  Select on the REVERSED index r = 0x3FFFFF - flat (flat < 2^22; larger
    // r = smaller index): the need_eq largest r.
    zero_hist(2048);
    scan_values(src, n, [&](int i, float v) { if (key_of(v) == kth) atomicAdd(&hist[(0x3FFFFFu - flat_of(i)) >> 11], 1u); });
    __syncthreads();
    pick_digit<2048>(hist, need_eq, ws, pick);
    const unsigned e0 = pick[0];
    const unsigned want = need_eq - pick[1];
    __syncthreads();
    zero_hist(2048);
    scan_values(src, n, [&](int i, float v) {
      if (key_of(v) == kth) {
        const unsigned r = 0x3FFFFFu - flat_of(i);
        if ((r >> 11) == e0) atomicAdd(&hist[r & 2047u], 1u);
      }
    });
    __syncthreads();
    pick_digit<2048>(hist, want, ws, pick);
    idx_thr = 0x3FFFFFu - ((e0 << 11) | pick[0]);        // flat indices are unique: exactly `want` ties of this group have r >= the picked r
    __syncthreads();
  }
  if (tid == 0) counter = 0u;
  __syncthreads();
  if constexpr (!MERGE) {
    // ---- the winners of the slice -> its scratch list (arbitrary order; unused entries read as empty)
    float* cv = p.cval + (long)b * p.ctot + p.seg_coff[seg];
    int* ci = p.cidx + (long)b * p.ctot + p.seg_coff[seg];
    scan_values(src, n, [&](int i, float v) {
      const unsigned k = key_of(v);
      if (k > kth || (k == kth && k != 0u && (unsigned)(start + i) <= idx_thr)) {
        const unsigned slot = atomicAdd(&counter, 1u);
        if (slot < (unsigned)K) { cv[slot] = v; ci[slot] = start + i; }
      }
    });
    __syncthreads();
    for (int s_ = (int)min(counter, (unsigned)K) + tid; s_ < K; s_ += PS_NT) { cv[s_] = -1.f; ci[s_] = 0; }
  } else {
    // ---- the winners of the level -> LDS, sorted by (value desc, flat asc), decoded into the level's slots in that order
    int npow = 2;
    while (npow < K) npow <<= 1;                         // K <= PS_KMAX (host)
    for (int i = tid; i < npow; i += PS_NT) sel[i] = 0ull;
    __syncthreads();
    scan_values(src, n, [&](int i, float v) {
      const unsigned k = key_of(v);
      if (k > kth || (k == kth && k != 0u && flat_of(i) <= idx_thr)) {
        const unsigned slot = atomicAdd(&counter, 1u);
        if (slot < (unsigned)K) sel[slot] = ((unsigned long long)k << 32) | (unsigned long long)(0x3FFFFFu - flat_of(i));
      }
    });
    __syncthreads();
    for (int size = 2; size <= npow; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < (npow >> 1); t += PS_NT) {
          const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
          const bool desc = ((lo & size) == 0);
          const unsigned long long a = sel[lo], c = sel[hi];
          if ((a < c) == desc) { sel[lo] = c; sel[hi] = a; }
        }
        __syncthreads();
      }
    const long row = (long)b * p.tot + p.off[lvl];
    const float* regs = p.reg[lvl] + (long)b * p.hw[lvl] * 4;
    const float* anc = p.anchors[lvl];
    const float W = p.im_wh[b * 2 + 0], H = p.im_wh[b * 2 + 1];
    const int L = p.L;
    for (int s_ = tid; s_ < K; s_ += PS_NT) {
      const unsigned long long e = sel[s_];
      const long o = row + s_;
      if (e == 0ull) {
        p.scores[o] = -1.f; p.labels[o] = 0; p.ids[o] = 0x7FFFFFFF;
        p.boxes[o * 4 + 0] = p.boxes[o * 4 + 1] = p.boxes[o * 4 + 2] = p.boxes[o * 4 + 3] = 0.f;
        continue;
      }
      const float v = __uint_as_float((unsigned)(e >> 32));
      const int i = (int)(0x3FFFFFu - (unsigned)(e & 0xFFFFFFFFull));
      const int loc = i / L, l = i - loc * L;
      const float* r = regs + (long)loc * 4;
      const float* a = anc + (long)loc * 4;
      const float w = a[2] - a[0] + 1.f, h = a[3] - a[1] + 1.f;
      const float cx = (a[2] + a[0]) * 0.5f, cy = (a[3] + a[1]) * 0.5f;
      const float lim = 4.135166556742356f;              // log(1000 / 16)
      const float dx = r[0] / 10.f, dy = r[1] / 10.f;
      const float dw = fminf(r[2] / 5.f, lim), dh = fminf(r[3] / 5.f, lim);
      const float pcx = dx * w + cx, pcy = dy * h + cy;
      const float pw = expf(dw) * w, ph = expf(dh) * h;
      p.boxes[o * 4 + 0] = fminf(fmaxf(pcx - 0.5f * (pw - 1.f), 0.f), W - 1.f);
      p.boxes[o * 4 + 1] = fminf(fmaxf(pcy - 0.5f * (ph - 1.f), 0.f), H - 1.f);
      p.boxes[o * 4 + 2] = fminf(fmaxf(pcx + 0.5f * (pw - 1.f), 0.f), W - 1.f);
      p.boxes[o * 4 + 3] = fminf(fmaxf(pcy + 0.5f * (ph - 1.f), 0.f), H - 1.f);
      p.scores[o] = sqrtf(v);
      p.labels[o] = p.label_ids[(long)b * p.lab_bs + l];
      p.ids[o] = p.idbase[lvl] + i;
    }
  }
}

namespace {
// host: cut the levels into slices; returns the number of scratch entries per image (or -1)
long post_plan(PostSelectParams& p, const int* hw, const int* k, int NL, int L) {
  int nseg = 0, off = 0;
  long idb = 0, ctot = 0;
  for (int l = 0; l < NL; ++l) {
    const long n = (long)hw[l] * L;
    if (n <= 0 || n >= (1L << 22) || k[l] <= 0 || k[l] > n || k[l] > PS_KMAX) return -1;
    p.hw[l] = hw[l]; p.k[l] = k[l]; p.off[l] = off; p.idbase[l] = (int)idb;
    p.coff[l] = (int)ctot;
    int cn = 0;
    for (long s0 = 0; s0 < n; s0 += PS_SEGLEN) {
      if (nseg >= PS_MAXSEG) return -1;
      const int len = (int)((n - s0) < PS_SEGLEN ? (n - s0) : PS_SEGLEN);
      const int kk = k[l] < len ? k[l] : len;
      p.seg_lvl[nseg] = (short)l; p.seg_start[nseg] = (int)s0; p.seg_len[nseg] = len; p.seg_k[nseg] = kk; p.seg_coff[nseg] = (int)ctot;
      ctot += kk; cn += kk; ++nseg;
    }
    p.cn[l] = cn;
    off += k[l];
    idb += n;
  }
  if (idb >= 0x7FFFFFFFL) return -1;
  p.nseg = nseg; p.tot = off; p.ctot = (int)ctot; p.NL = NL; p.L = L;
  return ctot;
}
}  // namespace

// bytes of the scratch mq_post_select_fwd needs (value + flat index of every slice winner); -1: unsupported sizes (NL > 8, a level with
// hw * L >= 2^22, k[l] > 2048 or > hw[l] * L, more than 64 slices of 32768 scores in total)
extern "C" long mq_post_select_workspace_bytes(const int* hw, const int* k, int NL, int B, int L) {
  if (NL <= 0 || NL > PS_MAXLVL || L <= 0 || B <= 0) return -1;
  PostSelectParams p;
  const long ctot = post_plan(p, hw, k, NL, L);
  return ctot < 0 ? -1 : ctot * (long)B * 8;
}

// ranked / reg / anchors: HOST arrays of NL device pointers (level l: [B, hw[l], L] fp32 scores with candidates > 0, [B, hw[l], 4] fp32
// deltas, [hw[l], 4] anchors); hw / k: HOST ints per level (locations; candidates kept = min(PRE_NMS_TOP_N, hw * L)); label_ids [L]
// (lab_bs 0) or [B, L] int32; im_wh [B, 2]; workspace: mq_post_select_workspace_bytes.  Outputs, caller-allocated: boxes [B, tot, 4] /
// scores [B, tot] (-1 = empty slot) / labels / ids [B, tot] int32 with tot = sum(k); level l owns the slots [sum(k[:l]), sum(k[:l + 1])),
// sorted by (score descending, flat index ascending) inside them, empty slots last.  -1: see mq_post_select_workspace_bytes.
extern "C" int mq_post_select_fwd(const float* const* ranked, const float* const* reg, const float* const* anchors, const int* hw, const int* k,
                                  int NL, int B, int L, const int* label_ids, long lab_bs, const float* im_wh, void* workspace, float* boxes,
                                  float* scores, int* labels, int* ids, void* stream) {
  if (B <= 0 || NL <= 0) return 0;
  if (NL > PS_MAXLVL || L <= 0) return -1;
  PostSelectParams p;
  const long ctot = post_plan(p, hw, k, NL, L);
  if (ctot < 0) return -1;
  for (int l = 0; l < NL; ++l) { p.ranked[l] = ranked[l]; p.reg[l] = reg[l]; p.anchors[l] = anchors[l]; }
  p.cval = (float*)workspace; p.cidx = (int*)((float*)workspace + ctot * B);
  p.label_ids = label_ids; p.lab_bs = lab_bs; p.im_wh = im_wh; p.boxes = boxes; p.scores = scores; p.labels = labels; p.ids = ids;
  p.B = B;
  hipLaunchKernelGGL(post_select_kernel<false>, dim3((unsigned)p.nseg, (unsigned)B), dim3(PS_NT), 0, (hipStream_t)stream, p);
  MQ_CHECK_LAUNCH();
  hipLaunchKernelGGL(post_select_kernel<true>, dim3((unsigned)NL, (unsigned)B), dim3(PS_NT), 0, (hipStream_t)stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Merge the per-level lists of an image (each sorted by (score desc, id asc), empty slots last) into ONE list in the same order: the
// position of an element = its position in its own list + the number of elements of every other list that precede it, found by binary
// search over the scores staged in LDS (ids grow with the level: on equal scores a lower level precedes).  Then the rows are scattered.
struct PostMergeParams { int off[PS_MAXLVL + 1]; int NL, tot; };

__global__ __launch_bounds__(PS_NT) void post_merge_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                           const int* __restrict__ labels, float* __restrict__ boxes_o,
                                                           float* __restrict__ scores_o, int* __restrict__ labels_o, int* __restrict__ nvalid,
                                                           PostMergeParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_sort[];
  float* sc = (float*)smem_sort;                                     // [tot]
  __shared__ unsigned live;
  const int b = blockIdx.x, tid = threadIdx.x;
  const long row = (long)b * p.tot;
  if (tid == 0) live = 0u;
  for (int i = tid; i < p.tot; i += PS_NT) sc[i] = scores[row + i];
  __syncthreads();
  unsigned mine = 0;
  for (int i = tid; i < p.tot; i += PS_NT) {
    const float s = sc[i];
    int l = 0;
    while (l + 1 < p.NL && i >= p.off[l + 1]) ++l;
    int rank = i - p.off[l];
    for (int m = 0; m < p.NL; ++m) {
      if (m == l) continue;
      // number of entries of list m that precede (s, level l): score > s, or score == s when m < l -- lists are descending
      int lo = p.off[m], hi = p.off[m + 1];
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const float t = sc[mid];
        if (t > s || (t == s && m < l)) lo = mid + 1; else hi = mid;
      }
      rank += lo - p.off[m];
    }
    const long o = row + rank;
    if (s > 0.f) {
      ++mine;
      scores_o[o] = s;
      labels_o[o] = labels[row + i];
      *(float4_*)(boxes_o + o * 4) = *(const float4_*)(boxes + (row + i) * 4);
    } else {
      scores_o[o] = -1.f; labels_o[o] = 0;
      *(float4_*)(boxes_o + o * 4) = (float4_){0.f, 0.f, 0.f, 0.f};
    }
  }
  if (mine) atomicAdd(&live, mine);
  __syncthreads();
  if (tid == 0) nvalid[b] = (int)live;
}

// boxes [B, tot, 4] / scores [B, tot] (<= 0: empty) / labels [B, tot]: NL lists per image (list l = slots [off[l], off[l + 1]), HOST ints,
// off[NL] = tot), each sorted by (score desc, id asc) with its empty slots last -- what mq_post_select_fwd writes -> the same rows as ONE
// list in that order (equal scores: lower list first), empty rows last, + nvalid [B].  tot <= 16384 (-1 beyond), NL <= 8.
extern "C" int mq_post_sort_fwd(const float* boxes, const float* scores, const int* labels, const int* off, int NL, float* boxes_o,
                                float* scores_o, int* labels_o, int* nvalid, int B, int tot, void* stream) {
  if (B <= 0 || tot <= 0) return 0;
  if (tot > 2 * SORT_MAX || NL < 1 || NL > PS_MAXLVL || off[0] != 0 || off[NL] != tot) return -1;
  PostMergeParams p;
  for (int l = 0; l <= PS_MAXLVL; ++l) p.off[l] = off[l < NL ? l : NL];
  p.NL = NL; p.tot = tot;
  const size_t smem = (size_t)tot * sizeof(float);
  static MqOncePerDevice attr;
  if (attr.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)post_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * SORT_MAX * sizeof(float)));
    if (e != hipSuccess) return (int)e;
    attr.done();
  }
  hipLaunchKernelGGL(post_merge_kernel, dim3((unsigned)B), dim3(PS_NT), smem, (hipStream_t)stream, boxes, scores, labels, boxes_o, scores_o,
                     labels_o, nvalid, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Final selection after the NMS: rows are score-sorted, keep[] marks the survivors (mq_ml_nms_topk: the sweep stops after K2 of them).
// out[b] = the first K kept rows + the kept rows behind them whose score EQUALS the K-th one (at most K2 - K), packed
// (x1, y1, x2, y2, score, label); unused rows (0, 0, 0, 0, -1, 0).  counts[b] = live rows | (1 << 16) when every tie slot is taken by
// a tie (more may exist behind the sweep's stop: the reference would return them all).
__global__ __launch_bounds__(PS_NT) void post_finalize_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                              const int* __restrict__ labels, const unsigned char* __restrict__ keep,
                                                              float* __restrict__ out, int* __restrict__ counts, int tot, int K, int K2) {
  __shared__ unsigned ws[PS_NW + 1];
  __shared__ float kth_score;
  __shared__ unsigned n_out;
  const int b = blockIdx.x, tid = threadIdx.x;
  const long row = (long)b * tot;
  float* ob = out + (long)b * K2 * 6;
  for (int i = tid; i < K2 * 6; i += PS_NT) ob[i] = (i % 6 == 4) ? -1.f : 0.f;
  if (tid == 0) { kth_score = -2.f; n_out = 0u; }
  __syncthreads();
  // rank of every kept row among the kept rows: contiguous chunks per thread, in order
  const int per = (tot + PS_NT - 1) / PS_NT;
  const int i0 = min(tid * per, tot), i1 = min(i0 + per, tot);
  unsigned mine = 0;
  for (int i = i0; i < i1; ++i) mine += (keep[row + i] != 0 && scores[row + i] > 0.f);
  unsigned total;
  unsigned rank = block_excl_scan(mine, ws, &total);
  {
    unsigned r = rank;
    for (int i = i0; i < i1; ++i)
      if (keep[row + i] != 0 && scores[row + i] > 0.f) {
        if ((int)r == K - 1) kth_score = scores[row + i];
        ++r;
      }
  }
  __syncthreads();
  const float ks = kth_score;
  unsigned r = rank, wrote = 0;
  for (int i = i0; i < i1; ++i) {
    const float s = scores[row + i];
    if (keep[row + i] != 0 && s > 0.f) {
      if ((int)r < K || ((int)r < K2 && s == ks)) {
        float* o = ob + (long)r * 6;
        const float4_ bx = *(const float4_*)(boxes + (row + i) * 4);
        o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2]; o[3] = bx[3]; o[4] = s; o[5] = (float)labels[row + i];
        ++wrote;
      }
      ++r;
    }
  }
  if (wrote) atomicAdd(&n_out, wrote);
  __syncthreads();
  if (tid == 0) {
    const unsigned nlive = n_out;                        // ranks are contiguous from 0: the live rows are rows 0 .. nlive - 1
    const int overflow = (K > 0 && K < K2 && K2 < tot && nlive == (unsigned)K2 && ob[(long)(K2 - 1) * 6 + 4] == ks && ks > 0.f) ? 1 : 0;
    counts[b] = (int)nlive | (overflow << 16);
  }
}

// boxes / scores / labels [B, tot] score-sorted, keep [B, tot] uint8 -> out [B, K2, 6] fp32, counts [B] int32 (bit 16: tie overflow).
// 1 <= K <= K2 <= tot.
extern "C" int mq_post_finalize_fwd(const float* boxes, const float* scores, const int* labels, const unsigned char* keep, float* out,
                                    int* counts, int B, int tot, int K, int K2, void* stream) {
  if (B <= 0) return 0;
  if (tot <= 0 || K < 1 || K2 < K || K2 > tot) return -1;
  hipLaunchKernelGGL(post_finalize_kernel, dim3((unsigned)B), dim3(PS_NT), 0, (hipStream_t)stream, boxes, scores, labels, keep, out, counts, tot,
                     K, K2);
  MQ_CHECK_LAUNCH();
  return 0;
}

#endif  // MQ_BF16
