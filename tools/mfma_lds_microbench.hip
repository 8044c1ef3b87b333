// Micro-benchmark for the question DESIGN.md section 10 ends with: kernels whose MFMAs each take a FRESH A-fragment from LDS (weights
// in swin_mlp_kernel at T = 1, K fragments in the VLFuse kernels at QB = 1) look LDS-bound on paper -- 1 KB of ds_read_b128 per
// v_mfma_f32_16x16x32_f16, i.e. 8 LDS cycles at 128 B/clk/CU for 16 MFMA cycles on ONE of four SIMDs.  Would 32 x 32 x 16 tiles (the same
// 1 KB per MFMA, twice the flops) lift that bound on gfx950?  For both shapes: cycles per wave for a loop of MFMAs whose A operand
// comes (a) from registers, (b) from LDS with one ds_read_b128 per MFMA, software-pipelined one read ahead; with 1, 2 and 4 waves per SIMD.
// Prints cycles per MFMA and the TFLOP/s the CU-level rate extrapolates to.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_microbench.hip -o /tmp/mfma_lds && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4_ __attribute__((ext_vector_type(4)));
typedef float float16_ __attribute__((ext_vector_type(16)));

template <bool BIG, bool FROM_LDS>
__global__ void mfma_lds_kernel(int iters, float* out, long long* cycles) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[16 * 1024];           // 32 KB: 32 fragments-worth of rows per wave slot
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) lds[i] = (_Float16)((i % 97) * 0.01f);
  __syncthreads();
  half8 b;
  for (int j = 0; j < 8; ++j) b[j] = (_Float16)(j * 0.25f - lane * 0.01f);
  // conflict-free fragment addressing: 16 rows x 8 halfs per 16-lane group, row pitch 72 halfs (144 B)
  const _Float16* base = lds + ((wave & 3) * 2048) + (lane & 15) * 72 + (lane >> 4) * 8;
  float4_ acc4[8];
  float16_ acc16[4];
  for (int i = 0; i < 8; ++i) acc4[i] = (float4_){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = 0.f;
  half8 a = *(const half8*)base;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (BIG) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        half8 an = a;
        if constexpr (FROM_LDS) an = *(const half8*)(base + ((it * 4 + i + 1) & 7) * 1152);   // next fragment while this MFMA runs
        acc16[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc16[i], 0, 0, 0);
        a = an;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        half8 an = a;
        if constexpr (FROM_LDS) an = *(const half8*)(base + ((it * 8 + i + 1) & 7) * 1152);
        acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4[i], 0, 0, 0);
        a = an;
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int i = 0; i < 8; ++i) r += acc4[i][0];
  for (int i = 0; i < 4; ++i) r += acc16[i][0] + acc16[i][15];
  if (r == 12345.678f) out[0] = r;
  if (blockIdx.x == 0 && lane == 0) cycles[wave] = t1 - t0;
}

template <bool BIG, bool FROM_LDS>
static void run(const char* name, float* out, long long* cyc) {
  const int iters = 4000;
  for (int wps : {1, 2, 4}) {
    const int threads = 256 * wps;
    hipLaunchKernelGGL((mfma_lds_kernel<BIG, FROM_LDS>), dim3(256), dim3(threads), 0, 0, iters, out, cyc);
    (void)hipDeviceSynchronize();
    long long h[16];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
    const double mfmas = (double)iters * (BIG ? 4 : 8), flops = BIG ? 32768.0 : 16384.0;
    const double per = mx / mfmas;                                        // cycles per MFMA of one wave
    const double cu_flops_per_clk = flops * (threads / 64) / per;         // all waves of the workgroup (= of the CU) together
    printf("%-44s %d wave(s)/SIMD: %6.1f cycles / MFMA / wave -> %7.1f flop/clk/CU = %6.0f TFLOP/s at 256 CUs x 2.4 GHz\n", name, wps, per,
           cu_flops_per_clk, cu_flops_per_clk * 256 * 2.4e9 / 1e12);
  }
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&cyc, 16 * sizeof(long long));
  run<false, false>("16x16x32, A from registers", out, cyc);
  run<false, true>("16x16x32, A from LDS (1 b128 read per MFMA)", out, cyc);
  run<true, false>("32x32x16, A from registers", out, cyc);
  run<true, true>("32x32x16, A from LDS (1 b128 read per MFMA)", out, cyc);
  return 0;
}
