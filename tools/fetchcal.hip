// fetchcal -- calibrates rocprofv3 FETCH_SIZE for the two load shapes of the M4RM tile kernel on gfx950:
//   k_rows : each wave-instruction reads 256 contiguous bytes (4 B per lane)   [B rows]
//   k_bcast: each wave-instruction reads 4 distinct dwords, 16 lanes each      [A words]
// Both sweep a buffer far larger than the 256 MiB Infinity Cache exactly once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_rows(const unsigned *p, size_t ndw, unsigned *sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndw; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
  if (acc == 0x1234567) sink[0] = acc;
}
__global__ void k_bcast(const unsigned *p, size_t ndw, unsigned *sink) {  // one dword per 16 lanes, 2 KiB between groups
  unsigned acc = 0;
  const size_t groups = ndw / 512;
  for (size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 16; g < groups; g += (size_t)gridDim.x * blockDim.x / 16) acc ^= p[g * 512];
  if (acc == 0x1234567) sink[0] = acc;
}
int main() {
  const size_t bytes = (size_t)2 << 30, ndw = bytes / 4;
  unsigned *p, *sink;
  CK(hipMalloc(&p, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(p, 1, bytes));
  hipLaunchKernelGGL(k_rows, dim3(4096), dim3(256), 0, 0, p, ndw, sink);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_bcast, dim3(4096), dim3(256), 0, 0, p, ndw, sink);
  CK(hipDeviceSynchronize());
  printf("k_rows read %zu bytes (4 B/lane coalesced); k_bcast touched %zu dwords = %zu cache lines of 64 B (%zu bytes)\n", bytes, ndw / 512, ndw / 512, ndw / 512 * 64);
  return 0;
}
