// Does an 8-byte LDS read of a wave cost less when part of its lanes are switched off?  (development tool; decides whether the
// elimination update -- 33 of 64 lanes active on a full-rank block -- would gain from packing two rows into one wave's read)
//   hipcc --offload-arch=gfx950 -O3 -o lds_exec_bench lds_exec_bench.hip && ./lds_exec_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
__global__ __launch_bounds__(1024) void k(u64 *out, int active, int iters, int pair) {
  extern __shared__ u64 tab[];  // 16 x 16 x 64 words
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16 * 16 * 64; i += 1024) tab[i] = i * 0x9E3779B97F4A7C15ull;
  __syncthreads();
  u64 acc = 0;
  unsigned sel = tid * 2654435761u;
  // pair: lanes 0..31 and 32..63 read different entries (two rows in one read); else one entry for the wave
  const int word = pair ? (lane & 31) : lane;
  if (lane < active) {
    for (int it = 0; it < iters; ++it) {
      unsigned s = __builtin_amdgcn_readfirstlane(sel) + (pair ? (lane >> 5) * 7u : 0u);
#pragma unroll
      for (int g = 0; g < 16; ++g) acc ^= tab[(g * 16 + ((s >> (2 * g)) & 15)) * 64 + word];
      sel = sel * 1664525u + 1013904223u + (unsigned)acc;
    }
  }
  out[blockIdx.x * 1024 + tid] = acc;
}
int main() {
  u64 *out;
  hipMalloc(&out, 256 * 1024 * 8);
  const int lds = 16 * 16 * 64 * 8;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 2000;
  for (int pair = 0; pair < 2; ++pair)
    for (int active : {64, 48, 33, 32, 16, 8}) {
      hipLaunchKernelGGL(k, dim3(256), dim3(1024), lds, 0, out, active, iters, pair);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(1024), lds, 0, out, active, iters, pair);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      // reads per CU: 16 waves x iters x 16
      const double reads = 16.0 * iters * 16;
      printf("pair=%d active lanes %2d: %.3f ms  = %.2f ns per wave-read per CU (at 2.2 GHz: %.1f clk)\n", pair, active, ms, ms * 1e6 / reads,
             ms * 1e6 / reads * 2.2);
    }
  return 0;
}
