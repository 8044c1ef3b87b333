// mq_window_attn_fwd: Swin (shifted-)window multi-head self-attention for gfx950.
//
// Replaces, in one launch, the reference chain of swint.py:201-234 + :111-139 -- F.pad to a multiple
// of the window, torch.roll(-shift), window_partition, q*scale @ k^T, + relative-position bias,
// + SW-MSA mask (-100 across regions), softmax, @ v, window_reverse, roll(+shift), crop -- none of
// which moves data here: the roll / pad / partition are folded into the load and store addresses.
//
//   qkv      : [B, H, W, 3*C] fp16 = Linear(norm1(x)) on the UNPADDED tokens, C = heads*32
//   qkv_bias : [3*C] fp16 -- q/k/v of a pad token (pad happens after norm1, so a pad token is an exact
//              zero vector and its qkv row equals the bias; pad tokens DO act as keys, swint.py quirk 7)
//   rel_bias : [heads, N, N] fp32 (relative_position_bias_table gathered by relative_position_index)
//   out      : [B, H, W, C] fp16 (attention output before `proj`)
// One wave per (window, head): N = ws*ws <= 64 tokens padded to 64, head_dim 32.
//   S (64x64)  = 16 x mfma 16x16x32 (K = head_dim = 32, one MFMA per 16x16 block), Q/K fragments are
//                read straight from global memory (16 B per lane, each element is used once);
//   P.V        = 16 x mfma, V goes through a transposed LDS tile, P through a per-wave LDS tile.
#include "common.h"

struct WinParams {
  const half_t* qkv; const half_t* qkv_bias; const float* rel_bias; half_t* out;
  int B, H, W, C, heads, ws, shift, Hp, Wp, nWx, nWy;
  float scale;
};

__global__ __launch_bounds__(256) void window_attn_kernel(WinParams p) {
  constexpr int VS = 64 + 8, PS = 64 + 8;
  __shared__ __attribute__((aligned(16))) half_t Vs_all[4][32 * VS];
  __shared__ __attribute__((aligned(16))) half_t Ps_all[4][64 * PS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const long unit = (long)blockIdx.x * 4 + wave;          // (b, wy, wx, head)
  const long total = (long)p.B * p.nWy * p.nWx * p.heads;
  if (unit >= total) return;                               // wave-uniform; no block-level sync below
  const int head = unit % p.heads;
  long win = unit / p.heads;
  const int wx = win % p.nWx; win /= p.nWx;
  const int wy = win % p.nWy;
  const int b = win / p.nWy;
  const int N = p.ws * p.ws;
  half_t* Vs = Vs_all[wave];
  half_t* Ps = Ps_all[wave];

  // token i of this window -> pointer to its qkv row (or the bias row for pad tokens), region id
  auto tok_ptr = [&](int i, bool& real, int& y, int& x, int& region) -> const half_t* {
    int ii = min(i, N - 1);
    int ys = wy * p.ws + ii / p.ws, xs = wx * p.ws + ii % p.ws;       // coords in the shifted frame
    y = ys + p.shift; if (y >= p.Hp) y -= p.Hp;                       // shifted[ys] = x[(ys+shift) % Hp]
    x = xs + p.shift; if (x >= p.Wp) x -= p.Wp;
    int ry = ys < p.Hp - p.ws ? 0 : (ys < p.Hp - p.shift ? 1 : 2);
    int rx = xs < p.Wp - p.ws ? 0 : (xs < p.Wp - p.shift ? 1 : 2);
    region = ry * 3 + rx;
    real = (y < p.H) && (x < p.W);
    return real ? p.qkv + (((long)b * p.H + y) * p.W + x) * (3 * p.C) : p.qkv_bias;
  };

  // ---- V -> transposed LDS tile Vs[d][key]; keys >= N are zero
  for (int c = lane; c < 64 * 4; c += 64) {
    int key = c >> 2, ch = c & 3;
    half8 v = zero8();
    if (key < N) {
      bool real; int y, x, rg;
      const half_t* row = tok_ptr(key, real, y, x, rg);
      v = *(const half8*)(row + 2 * p.C + head * 32 + ch * 8);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) Vs[(ch * 8 + j) * VS + key] = v[j];
  }

  // ---- Q, K fragments (A: rows = queries, B: rows of K), token index = block*16 + l15
  half8 qf[4], kf[4];
  int region_q[4];          // region of the rows this lane owns in C layout is needed per (rb, r) below
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    bool real; int y, x, rg;
    const half_t* row = tok_ptr(blk * 16 + l15, real, y, x, rg);
    qf[blk] = *(const half8*)(row + head * 32 + lg * 8);
    kf[blk] = *(const half8*)(row + p.C + head * 32 + lg * 8);
    region_q[blk] = rg;     // region of token blk*16 + l15 (used as the KEY region: col = l15)
  }

  const float* rel = p.rel_bias + (long)head * N * N;
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) {
    float4_ s[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) s[nb] = mfma16(qf[rb], kf[nb], (float4_){0.f, 0.f, 0.f, 0.f});
    // C layout: row i = rb*16 + lg*4 + r (query), col j = nb*16 + l15 (key)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int i = rb * 16 + lg * 4 + r;
      int ic = min(i, N - 1);
      int region_i;
      {
        int ys = wy * p.ws + ic / p.ws, xs = wx * p.ws + ic % p.ws;
        int ry = ys < p.Hp - p.ws ? 0 : (ys < p.Hp - p.shift ? 1 : 2);
        int rx = xs < p.Wp - p.ws ? 0 : (xs < p.Wp - p.shift ? 1 : 2);
        region_i = ry * 3 + rx;
      }
      float mx = MQ_NEG_BIG;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        int j = nb * 16 + l15;
        float v = MQ_NEG_BIG;
        if (j < N) {
          v = s[nb][r] * p.scale + rel[ic * N + j];
          if (p.shift > 0 && region_i != region_q[nb]) v += -100.0f;
        }
        s[nb][r] = v;
        mx = fmaxf(mx, v);
      }
      mx = group16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        float e = __expf(s[nb][r] - mx);
        s[nb][r] = e;
        sum += e;
      }
      sum = group16_sum(sum);
      float inv = 1.f / sum;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) Ps[i * PS + nb * 16 + l15] = (half_t)(s[nb][r] * inv);
    }
  }

  wave_lds_fence();
  // ---- O = P V : rows rb*16.., cols (d) db*16.., K over 64 keys
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) {
    float4_ o[2] = {(float4_){0.f, 0.f, 0.f, 0.f}, (float4_){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      half8 pf = *(const half8*)(Ps + (rb * 16 + l15) * PS + kk * 32 + lg * 8);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        half8 vf = *(const half8*)(Vs + (db * 16 + l15) * VS + kk * 32 + lg * 8);
        o[db] = mfma16(pf, vf, o[db]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int i = rb * 16 + lg * 4 + r;
      if (i < N) {
        bool real; int y, x, rg;
        tok_ptr(i, real, y, x, rg);
        if (real) {
          half_t* dst = p.out + (((long)b * p.H + y) * p.W + x) * p.C + head * 32;
          dst[l15] = (half_t)o[0][r];
          dst[16 + l15] = (half_t)o[1][r];
        }
      }
    }
  }
}

extern "C" int mq_window_attn_fwd(const void* qkv, const void* qkv_bias, const float* rel_bias, void* out,
                                  int B, int H, int W, int C, int heads, int ws, int shift, void* stream) {
  if (B <= 0) return 0;
  if (C != heads * 32 || ws * ws > 64 || shift < 0 || shift >= ws) return -1;
  WinParams p;
  p.qkv = (const half_t*)qkv; p.qkv_bias = (const half_t*)qkv_bias; p.rel_bias = rel_bias; p.out = (half_t*)out;
  p.B = B; p.H = H; p.W = W; p.C = C; p.heads = heads; p.ws = ws; p.shift = shift;
  p.Hp = (H + ws - 1) / ws * ws; p.Wp = (W + ws - 1) / ws * ws;
  p.nWy = p.Hp / ws; p.nWx = p.Wp / ws;
  p.scale = 1.0f / sqrtf(32.0f);
  long total = (long)B * p.nWy * p.nWx * heads;
  hipLaunchKernelGGL(window_attn_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
  MQ_CHECK_LAUNCH();
  return 0;
}
