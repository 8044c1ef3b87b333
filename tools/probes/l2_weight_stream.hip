// Probe (round 5, call 16): how fast ONE workgroup of 8 waves can stream an L2-resident weight matrix into VGPRs, by the shape of the load
// instruction.  The fused text kernels (gcp_fused.hip, bert_attn.hip) load MFMA B fragments straight from the [N][K] row-major weights: one
// wave instruction = 16 rows x 64 B (PAT 0).  The alternative is weights re-packed once on the host in fragment order, one wave instruction =
// 1 KiB contiguous (PAT 1).  PAT 2 (8 rows x 128 B, whole cache lines of the row-major matrix; would need a lane permute to become fragments) separates
// "half lines" from "many rows".  Same bytes, same number of instructions, same loads in flight (two groups of 12 ahead of the consumer).
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_weight_stream l2_weight_stream.hip ; run: ./l2_weight_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
constexpr int N = 512, K = 768, NT = 512, NTL = 4, G = 3, NG = K / 32 / G;

template <int PAT, int DEPTH>
__global__ __launch_bounds__(NT, 2) void stream_kernel(const unsigned short* __restrict__ w, unsigned* __restrict__ out, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
  const unsigned short* base[NTL];
  for (int j = 0; j < NTL; ++j)
    base[j] = PAT == 0 ? w + (size_t)(wave * 64 + j * 16 + n) * K + q * 8
            : PAT == 1 ? w + ((size_t)(wave * NTL + j) * (K / 32) * 64 + lane) * 8
                       : w + (size_t)(wave * 64 + j * 16 + (lane >> 3)) * K + (lane & 7) * 8;
  constexpr int step = PAT == 1 ? 64 * 8 : 32;          // halfs per k-step of 32 (PAT 2: see OFS)
#define OFS(s) (PAT == 2 ? ((s) & 1) * 8 * K + ((s) >> 1) * 64 : (s) * step)
  u4 r[DEPTH][G][NTL];
  unsigned acc = 0;
  for (int rep = 0; rep < reps; ++rep) {
    int zero = 0;
    asm volatile("" : "+s"(zero));                       // the addresses are loop-invariant: keep the compiler from hoisting the loads out of the repeat loop
    const unsigned short* bs[NTL];
    for (int j = 0; j < NTL; ++j) bs[j] = base[j] + zero;
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < NTL; ++j) r[d][g][j] = *(const u4*)(bs[j] + OFS(d * G + g));
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
      if (grp + DEPTH - 1 < NG) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int j = 0; j < NTL; ++j) r[(grp + DEPTH - 1) % DEPTH][g][j] = *(const u4*)(bs[j] + OFS((grp + DEPTH - 1) * G + g));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < NTL; ++j) { u4 v = r[grp % DEPTH][g][j]; asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %2" : "+v"(acc) : "v"(v[0]), "v"(v[3])); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (acc == 0x12345678u) out[blockIdx.x * NT + threadIdx.x] = acc;
}

template <int PAT, int DEPTH>
static void run(const unsigned short* w, unsigned* out, int grid, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<PAT, DEPTH>), dim3(grid), dim3(NT), 0, 0, w, out, reps);
  (void)hipEventRecord(a, 0);
  const int L = 20;
  for (int i = 0; i < L; ++i) hipLaunchKernelGGL((stream_kernel<PAT, DEPTH>), dim3(grid), dim3(NT), 0, 0, w, out, reps);
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / L, bytes = (double)N * K * 2 * reps;
  printf("{\"pattern\": \"%s\", \"groups_in_flight\": %d, \"workgroups\": %d, \"us\": %.1f, \"GBs_per_workgroup\": %.1f, \"TBs_total\": %.2f}\n",
         PAT == 0 ? "fragment 16 rows x 64 B" : PAT == 1 ? "packed 1 KiB contiguous" : "8 rows x 128 B", DEPTH - 1, grid, us, bytes / us / 1e3, bytes * grid / us / 1e6);
}

int main() {
  unsigned short* w; unsigned* out;
  (void)hipMalloc(&w, (size_t)N * K * 2); (void)hipMalloc(&out, 4096 * NT * 4);
  std::vector<unsigned short> h((size_t)N * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(i * 2654435761u >> 16);
  (void)hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int grid : {72, 256, 576}) {
    run<0, 2>(w, out, grid, 3); run<1, 2>(w, out, grid, 3); run<2, 2>(w, out, grid, 3);
    run<0, 3>(w, out, grid, 3); run<1, 3>(w, out, grid, 3); run<2, 3>(w, out, grid, 3);
  }
  return 0;
}
