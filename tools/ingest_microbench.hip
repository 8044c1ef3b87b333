// Micro-benchmark: how many bytes per clock can ONE CU pull through the vector-memory path (global_load_dwordx4) on
// MI355X, as a function of where the data lives (L1 / L2 / Infinity Cache / HBM), the access shape (fully coalesced
// 1 KB per wave-instruction vs scattered 128-byte / 64-byte segments) and the number of loads in flight.
// Not part of the product; evidence for DESIGN.md / profiles/README.md ("per-CU ingest").
//   hipcc --offload-arch=gfx950 -O3 tools/ingest_microbench.hip -o tools/ingest_microbench && tools/ingest_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

// SEG = contiguous bytes per group of lanes (1024: whole wave contiguous; 128: 8 lanes per segment; 64: 4 lanes)
template <int U, int SEG>
__global__ void ingest_kernel(const uint4v* __restrict__ buf, unsigned mask16, int iters, unsigned* out) {
  const int lane = threadIdx.x & 63;
  const unsigned gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  unsigned h = gw * 2654435761u + 12345u;
  uint4v acc = {0, 0, 0, 0};
  constexpr int LPS = SEG / 16;                  // lanes per segment
  const unsigned sub = lane % LPS, seg = lane / LPS;
  for (int it = 0; it < iters; ++it) {
    uint4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      h = h * 1664525u + 1013904223u;
      unsigned base;                             // in 16-byte units
      if (SEG == 1024) base = ((h >> 4) * 64u + lane) & mask16;
      else {
        unsigned hs = (h >> 4) + seg * 2246822519u;      // every segment of the wave somewhere else
        base = ((hs * LPS) + sub) & mask16;
      }
      v[u] = buf[base];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int U, int SEG>
static void run(const char* where, const uint4v* buf, size_t bytes, int waves_per_cu, unsigned* out, int ncu) {
  const unsigned mask16 = (unsigned)(bytes / 16 - 1);
  const int iters = 2000 / U * 4;
  dim3 block(256), grid(ncu * waves_per_cu / 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((ingest_kernel<U, SEG>), grid, block, 0, 0, buf, mask16, iters / 4, out);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((ingest_kernel<U, SEG>), grid, block, 0, 0, buf, mask16, iters, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double total = (double)grid.x * 4 * iters * U * 1024.0;
  const double tbs = total / (ms * 1e-3) / 1e12;
  printf("%-6s seg=%4d B  loads/wave=%2d  waves/CU=%2d : %7.2f TB/s  = %6.1f B/clk/CU (2.4 GHz)   %.3f ms\n", where, SEG, U,
         waves_per_cu, tbs, tbs * 1e12 / ncu / 2.4e9, ms);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d MHz\n", prop.name, ncu, prop.clockRate / 1000);
  size_t big = (size_t)2 << 30;
  uint4v* buf;
  unsigned* out;
  if (hipMalloc(&buf, big) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 64);
  hipMemset(buf, 1, big);
  struct { const char* name; size_t bytes; } levels[] = {{"L1", 16 << 10}, {"L2", 2 << 20}, {"MALL", 128 << 20}, {"HBM", big}};
  for (auto& lv : levels) {
    for (int wpc : {4, 8, 16}) {
      run<4, 1024>(lv.name, buf, lv.bytes, wpc, out, ncu);
      run<8, 1024>(lv.name, buf, lv.bytes, wpc, out, ncu);
      run<16, 1024>(lv.name, buf, lv.bytes, wpc, out, ncu);
      run<8, 128>(lv.name, buf, lv.bytes, wpc, out, ncu);
      run<16, 128>(lv.name, buf, lv.bytes, wpc, out, ncu);
      run<8, 64>(lv.name, buf, lv.bytes, wpc, out, ncu);
    }
  }
  return 0;
}
