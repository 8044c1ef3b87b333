// TEST INFRASTRUCTURE ONLY -- never on the product path (mq_det_amd/ does not know this directory exists).
//
// A host stand-in for <hip/hip_runtime.h>: the kernel sources under mq_det_amd/csrc/ are compiled UNCHANGED (two textual rewrites in
// tests/simt/build_emu.py: `extern __shared__ T x[];` and `asm volatile(...)`) for x86-64 and executed lane by lane -- every HIP
// thread is a fiber, a wavefront is 64 fibers that meet at each cross-lane operation (MFMA, shuffles, ballot, LDS transpose read,
// wave barrier), a workgroup meets at __syncthreads().  Purpose: run the gfx950 kernels' index arithmetic, fragment layouts, LDS
// images and masking logic against the oracle on a machine without a GPU (tests/test_simt_kernels_cpu.py).  It says nothing
// about speed, and floating-point results differ from the device in the last bits (expf / rounding order inside an MFMA).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <algorithm>

#define MQ_SIMT_EMULATION 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorLaunchFailure = 719 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace simt {
struct Fiber;
extern Fiber* cur;                                   // the fiber that is executing
extern dim3 g_blockIdx, g_blockDim, g_gridDim;
extern int g_error;
const uint3_& tid();
int lane();
void* dyn_smem();
void wave_sync();                                    // all live lanes of the wave
void wave_sync_then(void (*fn)(void*), void* ctx);   // the same; the LAST lane to arrive runs fn(ctx) before anybody continues
float* wave_tile(int buf);                           // [256] fp32 result buffer of the current wave (per exchange buffer)
float* wave_tile32();                                // [1024] fp32 result buffer of the current wave (32 x 32 MFMA)
void block_sync();                                   // all live threads of the workgroup
uint64_t* xslot(int lane, int buf);                  // exchange slots of the current wave: [2][64][8] x 8 bytes
int next_buf();                                      // alternating buffer index per collective
long op_seq();                                       // sequence number of the collective next_buf() was just called for (>= 1)
typedef void (*BodyFn)(void*);
void launch(dim3 grid, dim3 block, size_t shmem, BodyFn fn, void* ctx);

template <class F>
inline void launch_fn(dim3 grid, dim3 block, size_t shmem, F&& f) {
  launch(grid, block, shmem, [](void* c) { (*static_cast<F*>(c))(); }, (void*)&f);
}

template <class T>
inline T shfl_idx(T v, int src) {
  static_assert(sizeof(T) <= 8, "shuffle of > 8 bytes");
  const int b = next_buf();
  uint64_t w = 0;
  memcpy(&w, &v, sizeof(T));
  *xslot(lane(), b) = w;
  wave_sync();
  w = *xslot(src & 63, b);
  T o;
  memcpy(&o, &w, sizeof(T));
  return o;
}
}  // namespace simt

#define threadIdx (simt::tid())
#define blockIdx (simt::g_blockIdx)
#define blockDim (simt::g_blockDim)
#define gridDim (simt::g_gridDim)
#define warpSize 64

inline hipError_t hipGetLastError() { int e = simt::g_error; simt::g_error = 0; return e; }
template <class K>
inline hipError_t hipFuncSetAttribute(K, hipFuncAttribute, int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
// a "chip" of 4 compute units: the pass / tail split of mq_swin_mlp2_fwd is reachable with a few hundred tokens
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  simt::launch_fn(dim3(grid), dim3(block), (size_t)(shmem), [&]() { kernel(__VA_ARGS__); })

inline void __syncthreads() { simt::block_sync(); }
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return simt::shfl_idx(v, simt::lane() ^ mask); }
template <class T> inline T __shfl(T v, int src, int width = 64) { (void)width; return simt::shfl_idx(v, src); }
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) { (void)width; int s = simt::lane() + (int)d; return simt::shfl_idx(v, s < 64 ? s : simt::lane()); }
// Participation is decided when the lanes meet, not when a lane later reads the exchange slots (another lane may have run on and
// finished by then): every participant tags its slot with the operation's sequence number.
inline unsigned long long __ballot(int pred) {
  const int b = simt::next_buf();
  const uint64_t tag = (uint64_t)simt::op_seq();
  uint64_t* s = simt::xslot(simt::lane(), b);
  s[0] = pred ? 1 : 0;
  s[1] = tag;
  simt::wave_sync();
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) {
    const uint64_t* o = simt::xslot(l, b);
    if (o[1] == tag && o[0]) m |= 1ull << l;
  }
  return m;
}
// atomics: the fibers of a block run one at a time between synchronisation points, so a plain read-modify-write is atomic here
template <class T> inline T atomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; }
template <class T> inline T atomicMax(T* p, T v) { const T o = *p; if (v > o) *p = v; return o; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) { return __ballot(!pred) == 0; }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __fdividef(float a, float b) { return a / b; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
using std::max;
using std::min;

// ---- gfx950 builtins
typedef _Float16 simt_half8 __attribute__((ext_vector_type(8)));
typedef __bf16 simt_bf16x8 __attribute__((ext_vector_type(8)));
typedef float simt_float4 __attribute__((ext_vector_type(4)));
typedef __fp16 simt_fp16x4 __attribute__((__vector_size__(8)));
typedef short simt_short4 __attribute__((__vector_size__(8)));

// v_mfma_f32_16x16x32_{f16,bf16}: A lane l = A[l & 15][8 (l >> 4) ..], B lane l = B[8 (l >> 4) ..][l & 15], D lane l = D[4 (l >> 4) + r][l & 15]
// The last lane to arrive multiplies the whole tile once (plain loops over the deposited fragments) into the wave's result buffer;
// every lane then picks up its four values.
template <class V8>
inline void simt_mfma_tile(void* ctx) {
  const int buf = (int)(intptr_t)ctx;
  float A[16][32], B[32][16];
  for (int l = 0; l < 64; ++l) {
    V8 a, b;
    const uint64_t* s = simt::xslot(l, buf);
    memcpy(&a, s, sizeof(V8));
    memcpy(&b, s + sizeof(V8) / 8, sizeof(V8));
    for (int j = 0; j < 8; ++j) {
      A[l & 15][8 * (l >> 4) + j] = (float)a[j];
      B[8 * (l >> 4) + j][l & 15] = (float)b[j];
    }
  }
  float* D = simt::wave_tile(buf);
  for (int i = 0; i < 16; ++i) {
    float acc[16];
    for (int n = 0; n < 16; ++n) acc[n] = 0.f;
    for (int k = 0; k < 32; ++k) {
      const float a = A[i][k];
      for (int n = 0; n < 16; ++n) acc[n] += a * B[k][n];
    }
    for (int n = 0; n < 16; ++n) D[i * 16 + n] = acc[n];
  }
}
template <class V8>
inline simt_float4 simt_mfma_16x16x32(V8 a, V8 b, simt_float4 c) {
  const int buf = simt::next_buf(), l = simt::lane();
  uint64_t* s = simt::xslot(l, buf);
  memcpy(s, &a, sizeof(V8));
  memcpy(s + sizeof(V8) / 8, &b, sizeof(V8));
  simt::wave_sync_then(&simt_mfma_tile<V8>, (void*)(intptr_t)buf);
  const float* D = simt::wave_tile(buf);
  const int col = l & 15, r0 = 4 * (l >> 4);
  for (int r = 0; r < 4; ++r) c[r] += D[(r0 + r) * 16 + col];
  return c;
}
// v_mfma_f32_32x32x16_{f16,bf16} (for the kernels of the next round; layout assumed from the 32x32x8 family, to be confirmed on the
// device with tools/mfma_layout_probe.hip): A lane l = A[l & 31][8 (l >> 5) ..], B lane l = B[8 (l >> 5) ..][l & 31],
// D lane l, register 4 i + j = D[8 i + 4 (l >> 5) + j][l & 31]
typedef float simt_float16 __attribute__((ext_vector_type(16)));
template <class V8>
inline void simt_mfma32_tile(void* ctx) {
  const int buf = (int)(intptr_t)ctx;
  float A[32][16], B[16][32];
  for (int l = 0; l < 64; ++l) {
    V8 a, b;
    const uint64_t* s = simt::xslot(l, buf);
    memcpy(&a, s, sizeof(V8));
    memcpy(&b, s + sizeof(V8) / 8, sizeof(V8));
    for (int j = 0; j < 8; ++j) {
      A[l & 31][8 * (l >> 5) + j] = (float)a[j];
      B[8 * (l >> 5) + j][l & 31] = (float)b[j];
    }
  }
  float* D = simt::wave_tile32();
  for (int i = 0; i < 32; ++i)
    for (int n = 0; n < 32; ++n) {
      float acc = 0.f;
      for (int k = 0; k < 16; ++k) acc += A[i][k] * B[k][n];
      D[i * 32 + n] = acc;
    }
}
template <class V8>
inline simt_float16 simt_mfma_32x32x16(V8 a, V8 b, simt_float16 c) {
  const int buf = simt::next_buf(), l = simt::lane();
  uint64_t* s = simt::xslot(l, buf);
  memcpy(s, &a, sizeof(V8));
  memcpy(s + sizeof(V8) / 8, &b, sizeof(V8));
  simt::wave_sync_then(&simt_mfma32_tile<V8>, (void*)(intptr_t)buf);
  const float* D = simt::wave_tile32();
  const int col = l & 31, h = l >> 5;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) c[4 * i + j] += D[(8 * i + 4 * h + j) * 32 + col];
  simt::wave_sync();                                    // the single 32x32 result buffer is free again only when every lane has read it
  return c;
}
// the element type of the operands is taken from the arguments: _Float16, __bf16, or -- the fp32-OPERAND build of the kernel sources
// (tests/simt/build_emu.py, entry points *_f32: every `half_t` is a float) -- float, multiplied exactly as given
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) simt_mfma_32x32x16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) simt_mfma_32x32x16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) simt_mfma_16x16x32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) simt_mfma_16x16x32((a), (b), (c))
// v_mfma_f32_16x16x4_f32 (the fp32-operand build, csrc/common.h mfma16 under MQ_F32): A lane l = A[l & 15][l >> 4], B lane l = B[l >> 4][l & 15],
// D as above; every product is an exact fp32 multiply, the four products of an output are added in k order into the accumulator
inline void simt_mfma4_tile(void* ctx) {
  const int buf = (int)(intptr_t)ctx;
  float A[16][4], B[4][16];
  for (int l = 0; l < 64; ++l) {
    float ab[2];
    memcpy(ab, simt::xslot(l, buf), 8);
    A[l & 15][l >> 4] = ab[0];
    B[l >> 4][l & 15] = ab[1];
  }
  float* D = simt::wave_tile(buf);
  for (int i = 0; i < 16; ++i)
    for (int n = 0; n < 16; ++n) {
      float acc = 0.f;
      for (int k = 0; k < 4; ++k) acc += A[i][k] * B[k][n];
      D[i * 16 + n] = acc;
    }
}
inline simt_float4 simt_mfma_16x16x4_f32(float a, float b, simt_float4 c) {
  const int buf = simt::next_buf(), l = simt::lane();
  float ab[2] = {a, b};
  memcpy(simt::xslot(l, buf), ab, 8);
  simt::wave_sync_then(&simt_mfma4_tile, (void*)(intptr_t)buf);
  const float* D = simt::wave_tile(buf);
  const int col = l & 15, r0 = 4 * (l >> 4);
  for (int r = 0; r < 4; ++r) c[r] += D[(r0 + r) * 16 + col];
  return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) simt_mfma_16x16x4_f32((a), (b), (c))
// mfma16 of the fp32-operand build (csrc/common.h): eight 16x16x4 steps, step j contracting the k values 8 g + j (g = 0 .. 3) -- here as ONE lane
// exchange (eight cost 8 x the fiber switches: the fp32 tests took 9 minutes), products and additions in the order of the eight device steps
#define MQ_SIMT_EMULATION 1
inline void simt_mfma_8x4_tile(void* ctx) {
  const int buf = (int)(intptr_t)ctx;
  float* AB = simt::wave_tile32();                   // A [16][32] then B [32][16]
  for (int l = 0; l < 64; ++l) {
    float a[8], b[8];
    const uint64_t* s = simt::xslot(l, buf);
    memcpy(a, s, 32);
    memcpy(b, s + 4, 32);
    for (int j = 0; j < 8; ++j) {
      AB[(l & 15) * 32 + 8 * (l >> 4) + j] = a[j];
      AB[512 + (8 * (l >> 4) + j) * 16 + (l & 15)] = b[j];
    }
  }
}
template <class V8>
inline simt_float4 simt_mfma_16x16x32_f32_8x4(V8 a, V8 b, simt_float4 c) {
  static_assert(sizeof(V8) == 32, "fp32-operand fragments");
  const int buf = simt::next_buf(), l = simt::lane();
  uint64_t* s = simt::xslot(l, buf);
  memcpy(s, &a, 32);
  memcpy(s + 4, &b, 32);
  simt::wave_sync_then(&simt_mfma_8x4_tile, (void*)(intptr_t)buf);
  const float* A = simt::wave_tile32();
  const float* B = A + 512;
  const int col = l & 15, r0 = 4 * (l >> 4);
  for (int j = 0; j < 8; ++j)                        // step j = one v_mfma_f32_16x16x4_f32: its four products as a group, then into the accumulator
    for (int r = 0; r < 4; ++r) {
      float acc = 0.f;
      for (int g = 0; g < 4; ++g) acc += A[(r0 + r) * 32 + 8 * g + j] * B[(8 * g + j) * 16 + col];
      c[r] += acc;
    }
  return c;                                          // (the buffer is refilled by the NEXT exchange's tile function, which runs once every lane has arrived there)
}
// mfma16 of the SPLIT-PRECISE build (csrc/common.h, round 6): every fp32 operand x = hi + lo / 2^11 with hi = fp16(x), lo = fp16((x - hi) 2^11);
// the device issues v_mfma_f32_16x16x32_f16 on (hi_a, hi_b) into the accumulator and on (hi_a, lo_b), (lo_a, hi_b) into a zeroed second one that is
// added scaled by 2^-11.  Here: ONE lane exchange of the fp32 fragments (simt_mfma_8x4_tile's buffer), the split and the three products per lane.
inline void simt_mfma_split_core(const float (&Ah)[16][32], const float (&Al)[16][32], const float (&Bh)[32][16], const float (&Bl)[32][16]) {
  float* D = simt::wave_tile32();                    // main [16][16], then cross [16][16]
  for (int i = 0; i < 16; ++i) {
    float m[16], x[16];
    for (int n = 0; n < 16; ++n) m[n] = x[n] = 0.f;
    for (int k = 0; k < 32; ++k) {
      const float ah = Ah[i][k], al = Al[i][k];
      for (int n = 0; n < 16; ++n) {
        m[n] += ah * Bh[k][n];
        x[n] += ah * Bl[k][n] + al * Bh[k][n];
      }
    }
    for (int n = 0; n < 16; ++n) { D[i * 16 + n] = m[n]; D[256 + i * 16 + n] = x[n]; }
  }
}
// (the LAST lane to arrive splits the deposited fp32 fragments once and multiplies the whole tile; every lane then picks up its four values)
inline void simt_mfma_split_tile(void* ctx) {
  const int buf = (int)(intptr_t)ctx;
  static float Ah[16][32], Al[16][32], Bh[32][16], Bl[32][16];      // (one OS thread runs every fiber: tests/simt/simt_runtime.cpp)
  for (int l = 0; l < 64; ++l) {
    float f[16];
    memcpy(f, simt::xslot(l, buf), 64);
    for (int j = 0; j < 8; ++j) {
      const _Float16 ha = (_Float16)f[j], hb = (_Float16)f[8 + j];
      Ah[l & 15][8 * (l >> 4) + j] = (float)ha;
      Al[l & 15][8 * (l >> 4) + j] = (float)(_Float16)((f[j] - (float)ha) * 2048.0f);
      Bh[8 * (l >> 4) + j][l & 15] = (float)hb;
      Bl[8 * (l >> 4) + j][l & 15] = (float)(_Float16)((f[8 + j] - (float)hb) * 2048.0f);
    }
  }
  simt_mfma_split_core(Ah, Al, Bh, Bl);
}
template <class V8>
inline simt_float4 simt_mfma_16x16x32_split(V8 a, V8 b, simt_float4 c) {
  static_assert(sizeof(V8) == 32, "fp32-operand fragments");
  const int buf = simt::next_buf(), l = simt::lane();
  uint64_t* s = simt::xslot(l, buf);
  memcpy(s, &a, 32);
  memcpy(s + 4, &b, 32);
  simt::wave_sync_then(&simt_mfma_split_tile, (void*)(intptr_t)buf);
  const float* D = simt::wave_tile32();              // (refilled by the NEXT exchange's tile function, which runs once every lane has arrived there)
  const int col = l & 15, r0 = 4 * (l >> 4);
  for (int r = 0; r < 4; ++r) c[r] = (c[r] + D[(r0 + r) * 16 + col]) + D[256 + (r0 + r) * 16 + col] * (1.0f / 2048.0f);
  return c;
}
// mfma16_split on fragments that are ALREADY split (planar hi / lo LDS tiles of the split-precise kernels, csrc/common.h): the device's three
// v_mfma_f32_16x16x32_f16 -- (hi, hi) into the accumulator, (hi, lo) + (lo, hi) into a zeroed second one that is added scaled by 2^-11
// (one lane exchange; the LAST lane to arrive converts the 4 x 64 fragments once and multiplies the tile -- main and cross products -- into the wave's
// 32 x 32 result buffer; per-lane conversion of its own rows / column was 7 x the fp16 -> fp32 conversions and slower than three exchanges)
inline void simt_mfma_split_frag_tile(void* ctx) {
  const int buf = (int)(intptr_t)ctx;
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  static float Ah[16][32], Al[16][32], Bh[32][16], Bl[32][16];      // (one OS thread runs every fiber: tests/simt/simt_runtime.cpp)
  for (int l = 0; l < 64; ++l) {
    h8 f[4];
    memcpy(f, simt::xslot(l, buf), 64);
    for (int j = 0; j < 8; ++j) {
      Ah[l & 15][8 * (l >> 4) + j] = (float)f[0][j];
      Al[l & 15][8 * (l >> 4) + j] = (float)f[1][j];
      Bh[8 * (l >> 4) + j][l & 15] = (float)f[2][j];
      Bl[8 * (l >> 4) + j][l & 15] = (float)f[3][j];
    }
  }
  simt_mfma_split_core(Ah, Al, Bh, Bl);
}
template <class H8>
inline simt_float4 simt_mfma_16x16x32_split_frag(H8 ahi, H8 alo, H8 bhi, H8 blo, simt_float4 c) {
  static_assert(sizeof(H8) == 16, "fp16 fragments");
  const int buf = simt::next_buf(), l = simt::lane();
  uint64_t* s = simt::xslot(l, buf);
  memcpy(s, &ahi, 16);
  memcpy(s + 2, &alo, 16);
  memcpy(s + 4, &bhi, 16);
  memcpy(s + 6, &blo, 16);
  simt::wave_sync_then(&simt_mfma_split_frag_tile, (void*)(intptr_t)buf);
  const float* D = simt::wave_tile32();              // (refilled by the NEXT exchange's tile function, which runs once every lane has arrived there)
  const int col = l & 15, r0 = 4 * (l >> 4);
  for (int r = 0; r < 4; ++r) c[r] = (c[r] + D[(r0 + r) * 16 + col]) + D[256 + (r0 + r) * 16 + col] * (1.0f / 2048.0f);
  return c;
}
inline int __lane_id() { return simt::lane(); }

// ds_read_b64_tr_b16: inside each 16-lane group the 16 x 4 block of 16-bit elements addressed by the lanes (lane i: row i >> 2,
// columns 4 (i & 3) .. + 3 of a [4][16] matrix) comes back transposed: lane i receives column i, rows 0..3
inline simt_fp16x4 simt_ds_read_tr16(uintptr_t addr) {
  const int buf = simt::next_buf(), l = simt::lane();
  *simt::xslot(l, buf) = (uint64_t)addr;
  simt::wave_sync();
  const int base = l & ~15, i = l & 15;
  simt_fp16x4 o;
  for (int j = 0; j < 4; ++j) {
    const uint16_t* src = (const uint16_t*)(uintptr_t)(*simt::xslot(base + 4 * j + (i >> 2), buf));
    uint16_t h = src[i & 3];
    memcpy((char*)&o + 2 * j, &h, 2);
  }
  return o;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4f16(p) simt_ds_read_tr16((uintptr_t)(p))
// the same exchange for elements of any size (fp32-operand build: 4-byte "halfs"): lane i of a 16-lane group passes the address of 4
// contiguous elements (row i >> 2, columns 4 (i & 3) .. + 3) and receives column i, rows 0..3
template <class V4>
inline V4 simt_ds_read_tr_elems(const void* p) {
  constexpr size_t E = sizeof(V4) / 4;
  const int buf = simt::next_buf(), l = simt::lane();
  *simt::xslot(l, buf) = (uint64_t)(uintptr_t)p;
  simt::wave_sync();
  const int base = l & ~15, i = l & 15;
  V4 o;
  for (int j = 0; j < 4; ++j) {
    const char* src = (const char*)(uintptr_t)(*simt::xslot(base + 4 * j + (i >> 2), buf));
    memcpy((char*)&o + E * j, src + E * (i & 3), E);
  }
  return o;
}
#define __builtin_amdgcn_wave_barrier() simt::wave_sync()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
// global_load_lds_dword{,x4}: every lane copies `size` bytes from ITS global address to  (wave-uniform LDS base) + lane * size
inline void simt_global_load_lds(const void* g, void* lds, unsigned size) { memcpy((char*)lds + (size_t)simt::lane() * size, g, size); }
// (the size argument must be a literal in the product source -- hipcc crashes on sizeof(half8) there --, so the fp32-operand build,
// whose "16-bit" fragments are twice as wide, scales it here)
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) simt_global_load_lds((const void*)(g), (void*)(l), (unsigned)(size))
// raw buffer resource (stride 0: base + byte range) and the MUBUF form of the LDS copy; a lane whose bytes are not inside [0, num) gets zeros,
// as the hardware's range check returns them
struct simt_buffer_rsrc { const char* base; unsigned num; };
typedef simt_buffer_rsrc __amdgpu_buffer_rsrc_t;
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) (simt_buffer_rsrc{(const char*)(p), (unsigned)(num)})
inline void simt_buffer_load_lds(simt_buffer_rsrc r, void* lds, unsigned size, unsigned off) {
  char* dst = (char*)lds + (size_t)simt::lane() * size;
  if ((unsigned long long)off + size <= r.num) memcpy(dst, r.base + off, size);
  else memset(dst, 0, size);
}
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(r, l, size, voff, soff, imm, aux) \
  simt_buffer_load_lds((r), (void*)(l), (unsigned)(size), (unsigned)(voff) + (unsigned)(soff) + (unsigned)(imm))
inline float simt_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
#define __builtin_amdgcn_fmed3f(a, b, c) simt_fmed3f((a), (b), (c))
#define __builtin_amdgcn_s_barrier() simt::block_sync()
#define SIMT_ASM(...) ((void)0)
