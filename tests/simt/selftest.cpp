// TEST INFRASTRUCTURE ONLY: unit kernels for the emulation itself (tests/test_simt_kernels_cpu.py::test_emulation_*).
#include <hip/hip_runtime.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// C[16][16] = A[16][32] . B[32][16] with ONE v_mfma_f32_16x16x32_f16, operands taken from memory in the documented fragment layout
// (A: lane l holds row l & 15, k = 8 (l >> 4) ..; B: lane l holds column l & 15, the same k; D: rows 4 (l >> 4) + r, column l & 15)
__global__ void st_mfma_kernel(const _Float16* A, const _Float16* Bt, float* C) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[r * 32 + 8 * g + j]; b[j] = Bt[r * 32 + 8 * g + j]; }   // Bt = B transposed: [16][32]
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int j = 0; j < 4; ++j) C[(4 * g + j) * 16 + r] = c[j];
}
extern "C" int simt_selftest_mfma(const void* A, const void* Bt, float* C) {
  hipLaunchKernelGGL(st_mfma_kernel, dim3(1), dim3(64), 0, nullptr, (const _Float16*)A, (const _Float16*)Bt, C);
  return hipGetLastError();
}

// C[32][32] = A[32][16] . B[16][32] with ONE v_mfma_f32_32x32x16_f16 in the layout the emulation assumes (see hip_runtime.h)
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void st_mfma32_kernel(const _Float16* A, const _Float16* Bt, float* C) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[r * 16 + 8 * h + j]; b[j] = Bt[r * 16 + 8 * h + j]; }   // Bt = B transposed: [32][16]
  f16v c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) C[(8 * i + 4 * h + j) * 32 + r] = c[4 * i + j];
}
extern "C" int simt_selftest_mfma32(const void* A, const void* Bt, float* C) {
  hipLaunchKernelGGL(st_mfma32_kernel, dim3(1), dim3(64), 0, nullptr, (const _Float16*)A, (const _Float16*)Bt, C);
  return hipGetLastError();
}

// cross-lane: out[l] = in[l ^ 1] + in[(l + 5) & 63];  ballot of (in[l] > 0) into out64
__global__ void st_shfl_kernel(const float* in, float* out, unsigned long long* out64) {
  const int l = threadIdx.x;
  const float v = in[l];
  out[l] = __shfl_xor(v, 1) + __shfl(v, (l + 5) & 63);
  const unsigned long long m = __ballot(v > 0.f);
  if (l == 0) *out64 = m;
}
extern "C" int simt_selftest_shfl(const float* in, float* out, unsigned long long* out64) {
  hipLaunchKernelGGL(st_shfl_kernel, dim3(1), dim3(64), 0, nullptr, in, out, out64);
  return hipGetLastError();
}

// two waves: wave 1 writes LDS, wave 0 reads it.  with_barrier = 0 is a data race: the value read depends on which wave runs first.
__global__ void st_race_kernel(int* out, int with_barrier) {
  __shared__ int box[64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w == 0) box[l] = -1;
  __syncthreads();
  if (w == 1) box[l] = l;
  if (with_barrier) __syncthreads();
  if (w == 0) out[l] = box[l];
}
extern "C" int simt_selftest_race(int* out, int with_barrier) {
  hipLaunchKernelGGL(st_race_kernel, dim3(1), dim3(128), 0, nullptr, out, with_barrier);
  return hipGetLastError();
}

// a meeting point not reached by all lanes: half of the wave waits in a shuffle, the other half at the workgroup barrier -- the
// emulation must report the launch as failed ("deadlock"), not hang
__global__ void st_deadlock_kernel(int* out) {
  int v = threadIdx.x;
  if (threadIdx.x < 32) v = __shfl_xor(v, 1);
  else __syncthreads();
  if (threadIdx.x < 32) __syncthreads();
  out[threadIdx.x] = v;
}
extern "C" int simt_selftest_deadlock(int* out) {
  hipLaunchKernelGGL(st_deadlock_kernel, dim3(1), dim3(64), 0, nullptr, out);
  return hipGetLastError();
}
