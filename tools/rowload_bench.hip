// rowload_bench -- does it matter how a wave reads 32-byte rows?  (development micro-benchmark)
//   mode 0: lane L loads its own row: two 16-byte loads at 32 L and 32 L + 16 (what the narrow / tall-skinny kernels do:
//           every instruction touches all 16 cache lines of the wave's 2 KiB, half of each)
//   mode 1: coalesced: instruction j loads 16 bytes at 16 (64 j + L) (8 whole lines per instruction)
// Each lane XOR-folds what it loaded and writes 8 bytes per row-equivalent (like C of an A*v product); cold = rotating buffers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ A, unsigned long long *__restrict__ C, long long rows) {
  const long long base = ((long long)blockIdx.x * 256 + threadIdx.x);
  const int lane = threadIdx.x & 63;
  const long long wave0 = base - lane;  // first row of this wave
  if (base >= rows) return;
  uint4 a, b;
  if (MODE == 0) {
    a = A[2 * base];
    b = A[2 * base + 1];
  } else {
    a = A[2 * wave0 + lane];
    b = A[2 * wave0 + 64 + lane];
  }
  C[base] = (unsigned long long)(a.x ^ a.y ^ a.z ^ a.w) | ((unsigned long long)(b.x ^ b.y ^ b.z ^ b.w) << 32);
}
int main() {
  const long long rows = 1 << 20;
  const int nbuf = 12;
  std::vector<uint4 *> A(nbuf);
  std::vector<unsigned long long *> C(nbuf);
  for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&A[i], rows * 32)); CK(hipMalloc(&C[i], rows * 8)); CK(hipMemset(A[i], i + 1, rows * 32)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode)
    for (int cold = 0; cold < 2; ++cold) {
      const int kk = cold ? nbuf : 1, reps = 240;
      for (int w = 0; w < 2; ++w) {
        if (w) CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) {
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(rows / 256), dim3(256), 0, 0, A[r % kk], C[r % kk], rows);
          else hipLaunchKernelGGL(k<1>, dim3(rows / 256), dim3(256), 0, 0, A[r % kk], C[r % kk], rows);
        }
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("mode %d (%s) %s: %.2f us per launch, %.0f GB/s\n", mode, mode ? "coalesced 16 B/lane" : "row per lane, 2 x 16 B", cold ? "cold" : "warm",
             ms * 1e3 / reps, rows * 40.0 / (ms * 1e-3 / reps) / 1e9);
    }
  return 0;
}
