// Device probe of the lane layouts the kernel-source emulation (tests/simt/include/hip/hip_runtime.h) assumes:
//   v_mfma_f32_16x16x32_f16, v_mfma_f32_32x32x16_f16 (assumed from the 32x32x8 family -- not used by any shipped kernel yet) and
//   ds_read_b64_tr_b16.  Each instruction runs once on known operands; the host compares with a plain matmul / transpose under the
//   assumed layout and prints PASS / FAIL.     hipcc --offload-arch=gfx950 -O2 tools/mfma_layout_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(8)));

__global__ void probe16(const _Float16* A, const _Float16* Bt, float* C) {       // A [16][32], Bt = B^T [16][32], C [16][16]
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[r * 32 + 8 * g + j]; b[j] = Bt[r * 32 + 8 * g + j]; }
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int j = 0; j < 4; ++j) C[(4 * g + j) * 16 + r] = c[j];
}
__global__ void probe32(const _Float16* A, const _Float16* Bt, float* C) {       // A [32][16], Bt = B^T [32][16], C [32][32]
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[r * 16 + 8 * h + j]; b[j] = Bt[r * 16 + 8 * h + j]; }
  f16v c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) C[(8 * i + 4 * h + j) * 32 + r] = c[4 * i + j];
}
// per 16-lane group a [4][16] block: lane i passes the address of row i >> 2, columns 4 (i & 3) ..; assumed result: column i, rows 0..3
__global__ void probe_tr(const _Float16* M, _Float16* out) {                        // M [4 groups][4][16] -> out [64][4]
  __shared__ __attribute__((aligned(16))) _Float16 tile[4 * 4 * 16];
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) tile[i] = M[i];
  __syncthreads();
  const int grp = l >> 4, i = l & 15;
  const _Float16* p = tile + grp * 64 + (i >> 2) * 16 + (i & 3) * 4;
  fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)p);
  h4 o;
  __builtin_memcpy(&o, &v, 8);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = o[j];
}

int main() {
  std::vector<_Float16> A(512), Bt(512), M(256);
  for (int i = 0; i < 512; ++i) { A[i] = (_Float16)(((i * 37) % 17 - 8) * 0.125f); Bt[i] = (_Float16)(((i * 53) % 13 - 6) * 0.25f); }
  for (int i = 0; i < 256; ++i) M[i] = (_Float16)i;
  _Float16 *dA, *dB, *dM, *dT;
  float* dC;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dM, 512); hipMalloc(&dT, 512); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(dM, M.data(), 512, hipMemcpyHostToDevice);
  std::vector<float> C(1024);
  int bad = 0;
  hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) {
    float r = 0; for (int k = 0; k < 32; ++k) r += (float)A[i * 32 + k] * (float)Bt[n * 32 + k];
    bad += std::fabs(r - C[i * 16 + n]) > 1e-3f;
  }
  printf("%s v_mfma_f32_16x16x32_f16 layout (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
  int bad32 = 0;
  hipLaunchKernelGGL(probe32, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) {
    float r = 0; for (int k = 0; k < 16; ++k) r += (float)A[i * 16 + k] * (float)Bt[n * 16 + k];
    bad32 += std::fabs(r - C[i * 32 + n]) > 1e-3f;
  }
  printf("%s v_mfma_f32_32x32x16_f16 layout (%d mismatches)\n", bad32 ? "FAIL" : "PASS", bad32);
  std::vector<_Float16> T(256);
  hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, dM, dT);
  hipMemcpy(T.data(), dT, 512, hipMemcpyDeviceToHost);
  int badt = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) badt += (float)T[l * 4 + j] != (float)M[(l >> 4) * 64 + j * 16 + (l & 15)];
  printf("%s ds_read_b64_tr_b16 semantics (%d mismatches)\n", badt ? "FAIL" : "PASS", badt);
  return (bad || bad32 || badt) ? 1 : 0;
}
