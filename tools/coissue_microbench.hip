// Micro-benchmark for the open question of profiles/README.md (DCNv2 section): when the two waves of a SIMD run
// DIFFERENT instruction classes -- one a stream of MFMAs, the other a stream of independent VALU FMAs (or LDS reads) --
// do they overlap, or does the SIMD serialise them?  Prints cycles per wave for each mix; "overlap" means the mixed
// case costs ~max(a, b), "serialised" ~a + b.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_microbench.hip -o tools/coissue_microbench && tools/coissue_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4_ __attribute__((ext_vector_type(4)));

// mode bit 0: even waves run MFMAs; bit 1: odd waves run VALU FMAs; bit 2: odd waves run LDS reads instead
// waves_per_simd = blockDim.x / 256 (waves go to SIMDs in cyclic order, so wave w and w + 4 share a SIMD)
__global__ void coissue_kernel(int mode, int iters, float* out, long long* cycles) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  const bool first_group = wave < (blockDim.x >> 7);          // lower half of the waves
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(lane * 0.01f + j); b[j] = (_Float16)(j * 0.5f - lane * 0.02f); }
  float4_ acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (float4_){0.f, 0.f, 0.f, 0.f};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = lane * 0.001f + i;
  const long long t0 = __builtin_readcyclecounter();
  if (first_group && (mode & 1)) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
  } else if (!first_group && (mode & 2)) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);      // 32 independent-ish VALU FMAs
    }
  } else if (!first_group && (mode & 4)) {
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) s += lds[(lane * 4 + i * 256 + it) & 4095];
    }
    v[0] += s;
  }
  const long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
  for (int i = 0; i < 16; ++i) r += v[i];
  if (r == 12345.678f) out[0] = r;
  if (blockIdx.x == 0 && lane == 0) cycles[wave] = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 64); hipMalloc(&cyc, 16 * sizeof(long long));
  const int iters = 2000;
  struct { const char* name; int mode; } cases[] = {
      {"MFMA only (8 x 16x16x32 per iter)", 1}, {"VALU only (32 fma per iter)", 2}, {"LDS only (8 ds_read per iter)", 4},
      {"MFMA || VALU", 3}, {"MFMA || LDS", 5}};
  for (int waves = 8; waves <= 16; waves += 8)
    for (auto& c : cases) {
      hipLaunchKernelGGL(coissue_kernel, dim3(256), dim3(64 * waves), 0, 0, c.mode, iters, out, cyc);
      hipLaunchKernelGGL(coissue_kernel, dim3(256), dim3(64 * waves), 0, 0, c.mode, iters, out, cyc);
      hipDeviceSynchronize();
      long long h[16];
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      printf("%2d waves/WG  %-36s : wave0 (MFMA group) %8.1f cyc/iter   wave%d (other group) %8.1f cyc/iter\n", waves, c.name,
             (double)h[0] / iters, waves / 2, (double)h[waves / 2] / iters);
    }
  return 0;
}
