// mq_align_fused_fwd: the prediction heads of VLDyHead and the per-location half of ATSS post-processing as ONE kernel for gfx950.
//
// Reference: rpn/vldyhead.py:853-888 -- bbox_pred / centerness 1x1 convs on the tower output (+ per-level Scale), dot-product
// alignment  logits = x . (proj_tokens / exp(log_scale))^T + bias,  clamp +-50000 -- followed by rpn/inference.py:656-683 and
// :772-824: sigmoid, token -> class aggregation (MEAN / MAX / POWER; ONEHOT through the host-built index), threshold, x
// sigmoid(centerness).  Round 2 ran this as a library bmm that wrote the [B, 22400, 256] logits to HBM in fp16 (92 MB at B = 8), five
// small GEMMs for the box / centerness rows and five launches of align_scores_kernel that read the logits straight back.
// Here a workgroup owns 128 pyramid tokens of one image (8 waves x 16 rows, all levels in one launch):
//   * the text operand -- the live rows of tk [T, 256] -- is staged once in LDS (row pitch 264 halfs: conflict-free b128 fragment
//     reads); token rows come straight from HBM as A fragments; the 5 box / centerness weight rows are one more 16-column block;
//   * v_mfma_f32_16x16x32: logits of 16 tokens x 16 text tokens per step, fp32 accumulators for all live text columns at once;
//   * epilogue in registers / LDS: + bias, clamp, sigmoid -> the wave's [16, T] probabilities overwrite the (no longer needed) text
//     operand in LDS; lanes then aggregate each (token, label) pair over the label's token positions, threshold, multiply by
//     sigmoid(centerness) and write the ranked score; box deltas get their level's Scale (fp32 out).
// The logits never reach HBM (optional fp32 debug output for the parity ladder).  Outputs are LEVEL-MAJOR (level l: [B, HW_l, ...]
// contiguous at token offset lvl_off[l] * B) so that the per-level top-k / box decode downstream read contiguous tensors.
// Algorithmic HBM bytes per token: 512 (features) + 4 L (scores) + 8 (box) [+ 4 L class scores] vs 512 + 2 * 512 + ... before.
#include "common.h"

MQ_NAMESPACE_BEGIN

#define AF_MAXLVL 8
struct AlignFusedParams {
  const half_t* tok;        // [B, N, 256]
  const half_t* tk;         // [B, T, 256]   projected text tokens / exp(log_scale), 16-bit
  const float* tbias;       // [B, T]
  const half_t* wbc;        // [16, 256]     rows 0-3 bbox_pred.weight, row 4 centerness.weight, rows 5-15 zero
  const float* bbc;         // [8]           their biases
  const float* scales;      // [NL]          per-level Scale of the box deltas
  const int* tokidx;        // [L, MT] (tok_bs = 0) or [B, L, MT] (tok_bs = L * MT)
  long tok_bs;
  float* ranked;            // level-major [sum_l B * HW_l * L]
  float* cls_out;           // same layout or NULL
  float* reg;               // level-major [sum_l B * HW_l * 4] box deltas, fp32 (a 16-bit store here was +55 % on the floor ratio of bbox_reg)
  float* ctr_out;           // [B, N] centerness logits
  float* logits;            // [B, N, T] fp32 dot products (without bias) or NULL
  int lvl_off[AF_MAXLVL + 1];
  int B, N, T, TL, L, MT, NL, agg;
  float thr;
};

constexpr int AF_C = 256, AF_KS = AF_C / 32, AF_PITCH = AF_C + 8, AF_NBMAX = 16, AF_NW = 8, AF_BM = 16 * AF_NW;

__global__ __launch_bounds__(64 * AF_NW) void align_fused_kernel(AlignFusedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* ts = (half_t*)smem;                                 // [TL][AF_PITCH]: live text rows
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = (p.N + AF_BM - 1) / AF_BM;
  const int b = blockIdx.x / tiles, tile = blockIdx.x - b * tiles;
  const int row0 = tile * AF_BM + wave * 16;                  // this wave's 16 tokens (rows >= N: clamped loads, no stores)
  const int nb = p.TL >> 4;                                   // live 16-column blocks of text tokens (<= AF_NBMAX)

  // ---- token rows -> A fragments (lane (l15, g): row l15, channels 32 ks + 8 g .. + 7), all 8 k-steps in flight
  half8 a[AF_KS];
  {
    const long r = min(row0 + l15, p.N - 1);
    const half_t* src = p.tok + ((long)b * p.N + r) * AF_C + g * 8;
#pragma unroll
    for (int ks = 0; ks < AF_KS; ++ks) a[ks] = *(const half8*)(src + ks * 32);
  }
  // ---- text operand -> LDS (16-byte pieces).  The 16 box / centerness weight rows (8 KB, shared by every workgroup: L2 / L1 hot) are
  // read as B fragments straight from memory instead: with them in the tile a 144-token caption needs 84.5 KB -- one workgroup per CU.
  {
    const int pieces = p.TL * (AF_C / 8);
    const half_t* tkb = p.tk + (long)b * p.T * AF_C;
    for (int i = tid; i < pieces; i += 64 * AF_NW) {
      const int r = i >> 5, c8 = i & 31;
      *(half8*)(ts + r * AF_PITCH + c8 * 8) = *(const half8*)(tkb + (long)min(r, p.T - 1) * AF_C + c8 * 8);
    }
  }
  half8 wb[AF_KS];
#pragma unroll
  for (int ks = 0; ks < AF_KS; ++ks) wb[ks] = *(const half8*)(p.wbc + l15 * AF_C + ks * 32 + g * 8);
  __syncthreads();

  // ---- logits: acc[cb] = rows x text columns 16 cb .. + 15; bc = the box / centerness block
  float4_ acc[AF_NBMAX];
#pragma unroll
  for (int cb = 0; cb < AF_NBMAX; ++cb) {
    acc[cb] = (float4_){0.f, 0.f, 0.f, 0.f};
    if (cb < nb) {
      const half_t* bp = ts + (cb * 16 + l15) * AF_PITCH + g * 8;
#pragma unroll
      for (int ks = 0; ks < AF_KS; ++ks) acc[cb] = mfma16(a[ks], *(const half8*)(bp + ks * 32), acc[cb]);
    }
  }
  float4_ bc = (float4_){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < AF_KS; ++ks) bc = mfma16(a[ks], wb[ks], bc);
  __syncthreads();                                            // every wave is done with the text operand: LDS is re-used below

  // lane holds logits[row = 4 g + r][col = 16 cb + l15]
  float* sw = (float*)smem + (long)wave * 16 * p.TL;          // [16][TL] probabilities of this wave's rows
  float* cw = (float*)smem + (long)AF_NW * 16 * p.TL + wave * 16;      // [16] sigmoid(centerness) of this wave's rows
  const float* tb = p.tbias + (long)b * p.T;
#pragma unroll
  for (int cb = 0; cb < AF_NBMAX; ++cb) {
    if (cb < nb) {
      const int col = cb * 16 + l15;
      const float bias = col < p.T ? tb[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float v = acc[cb][r];
        if (p.logits && row0 + row < p.N && col < p.T) p.logits[((long)b * p.N + row0 + row) * p.T + col] = v;
        v = fminf(fmaxf(v + bias, -50000.f), 50000.f);
        sw[row * p.TL + col] = 1.f / (1.f + __expf(-v));
      }
    }
  }
  // box deltas (columns 0-3, x level Scale) and centerness (column 4)
  if (l15 < 5) {
    const float bias = p.bbc[l15];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = row0 + 4 * g + r;
      if (n < p.N) {
        int lv = 0;
#pragma unroll
        for (int k = 1; k < AF_MAXLVL; ++k) lv += (k < p.NL && n >= p.lvl_off[k]);
        const float v = bc[r] + bias;
        if (l15 < 4) {
          const long hw = p.lvl_off[lv + 1] - p.lvl_off[lv];
          p.reg[((long)p.lvl_off[lv] * p.B + (long)b * hw + (n - p.lvl_off[lv])) * 4 + l15] = v * p.scales[lv];
        } else {
          p.ctr_out[(long)b * p.N + n] = v;
          cw[4 * g + r] = 1.f / (1.f + __expf(-v));
        }
      } else if (l15 == 4) {
        cw[4 * g + r] = 0.f;
      }
    }
  }
  wave_lds_fence();

  // ---- (token, label) pairs of this wave: aggregate the label's token probabilities, threshold, x sigmoid(centerness)
  const int* tix = p.tokidx + (long)b * p.tok_bs;
  const int pairs = 16 * p.L;
  for (int q = lane; q < pairs; q += 64) {
    const int row = q / p.L, l = q - row * p.L;
    const int n = row0 + row;
    if (n >= p.N) continue;
    float s = p.agg == 2 ? 1.f : 0.f;
    int cnt = 0;
    for (int j = 0; j < p.MT; ++j) {
      const int t = tix[l * p.MT + j];
      if (t >= 0) {
        // a token beyond the live columns cannot occur for a positive_map built from the caption (t < kv_len <= TL); guard anyway
        const float v = t < p.TL ? sw[row * p.TL + t] : 0.f;
        s = p.agg == 0 ? s + v : (p.agg == 1 ? fmaxf(s, v) : s * v);
        ++cnt;
      }
    }
    float cls = 0.f;                                          // a label without tokens scores 0 (never a candidate)
    if (cnt > 0) cls = p.agg == 0 ? s / (float)cnt : (p.agg == 1 ? s : powf(s, 1.f / (float)cnt));
    int lv = 0;
#pragma unroll
    for (int k = 1; k < AF_MAXLVL; ++k) lv += (k < p.NL && n >= p.lvl_off[k]);
    const long hw = p.lvl_off[lv + 1] - p.lvl_off[lv];
    const long o = ((long)p.lvl_off[lv] * p.B + (long)b * hw + (n - p.lvl_off[lv])) * p.L + l;
    if (p.cls_out) p.cls_out[o] = cls;
    // candidates are decided by the class score alone (rpn/inference.py:677): keep them > 0 even if the product with a vanishing
    // centerness underflows, so that "value > 0" identifies a candidate downstream
    p.ranked[o] = cls > p.thr ? fmaxf(cls * cw[row], 1.17549435e-38f) : -1.f;
  }
}

// tok [B, N, 256], tk [B, T, 256] 16-bit; tbias [B, T] fp32; wbc [16, 256] 16-bit, bbc [8] / scales [NL] fp32; tokidx [L, MT] (tok_bs 0) or
// [B, L, MT] (tok_bs = L * MT) int32; lvl_off [NL + 1] HOST ints (token offsets of the levels, lvl_off[NL] = N); kv_max: upper bound of the
// live text tokens (0 = T): text columns >= 16 ceil(kv_max / 16) are never scored.  Outputs (caller-allocated): ranked / cls_out (or NULL)
// level-major fp32 [sum_l B HW_l L], reg level-major fp32 [sum_l B HW_l 4], ctr_out [B, N] fp32, logits [B, N, T] fp32 or NULL.
// agg: 0 MEAN, 1 MAX, 2 POWER.  Returns -1 for unsupported shapes (T > 256, NL > 8, L * MT == 0).
extern "C" int MQ_SYM(mq_align_fused_fwd)(const void* tok, const void* tk, const float* tbias, const void* wbc, const float* bbc,
                                  const float* scales, const int* tokidx, long tok_bs, const int* lvl_off, float* ranked, float* cls_out,
                                  float* reg, float* ctr_out, float* logits, int B, int N, int T, int kv_max, int L, int MT, int NL,
                                  float thr, int agg, void* stream) {
  if (B <= 0 || N <= 0) return 0;
  if (T <= 0 || T > 16 * AF_NBMAX || NL < 1 || NL > AF_MAXLVL || L <= 0 || MT <= 0 || agg < 0 || agg > 2) return -1;
  AlignFusedParams p;
  p.tok = (const half_t*)tok; p.tk = (const half_t*)tk; p.tbias = tbias; p.wbc = (const half_t*)wbc; p.bbc = bbc; p.scales = scales;
  p.tokidx = tokidx; p.tok_bs = tok_bs; p.ranked = ranked; p.cls_out = cls_out; p.reg = reg; p.ctr_out = ctr_out; p.logits = logits;
  for (int i = 0; i <= AF_MAXLVL; ++i) p.lvl_off[i] = lvl_off[i < NL ? i : NL];
  if (p.lvl_off[0] != 0 || p.lvl_off[NL] != N) return -1;
  const int live = (kv_max > 0 && kv_max < T) ? kv_max : T;
  p.B = B; p.N = N; p.T = T; p.TL = (live + 15) / 16 * 16; p.L = L; p.MT = MT; p.NL = NL; p.agg = agg; p.thr = thr;
  const size_t text = (size_t)p.TL * AF_PITCH * sizeof(half_t);
  const size_t prob = ((size_t)AF_NW * 16 * p.TL + AF_NW * 16) * sizeof(float);
  const size_t smem = text > prob ? text : prob;
  static MqOncePerDevice attr;
  if (attr.first()) {
    hipError_t e = hipFuncSetAttribute((const void*)align_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr.done();
  }
  const unsigned grid = (unsigned)(B * ((N + AF_BM - 1) / AF_BM));
  hipLaunchKernelGGL(align_fused_kernel, dim3(grid), dim3(64 * AF_NW), smem, (hipStream_t)stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}

MQ_NAMESPACE_END
