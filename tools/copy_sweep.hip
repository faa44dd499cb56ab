// copy_sweep -- how fast can a read + write stream go on this chip?  Variants of a grid-stride copy (development tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// U loads of 16 bytes in flight per thread; MODE 0 plain, 1 non-temporal loads and stores, 2 non-temporal stores only;
// CONTIG 1: a workgroup walks a contiguous slab (its U pieces are adjacent 16-KiB rows), 0: grid-stride
template <int U, int MODE, int CONTIG>
__global__ __launch_bounds__(1024) void copyk(u32x4 *__restrict__ dst, const u32x4 *__restrict__ src, long long n16) {
  const long long nthreads = (long long)gridDim.x * blockDim.x;
  long long i0, step, inner;
  if (CONTIG) {
    const long long per_wg = (n16 + gridDim.x - 1) / gridDim.x;
    i0 = (long long)blockIdx.x * per_wg + threadIdx.x;
    inner = blockDim.x;
    step = (long long)U * blockDim.x;
    const long long end = min(n16, (long long)(blockIdx.x + 1) * per_wg);
    for (long long i = i0; i < end; i += step) {
      u32x4 v[U];
#pragma unroll
      for (int k = 0; k < U; ++k)
        if (i + k * inner < end) v[k] = MODE == 1 ? __builtin_nontemporal_load(src + i + k * inner) : src[i + k * inner];
#pragma unroll
      for (int k = 0; k < U; ++k)
        if (i + k * inner < end) {
          if (MODE) __builtin_nontemporal_store(v[k], dst + i + k * inner);
          else dst[i + k * inner] = v[k];
        }
    }
  } else {
    i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    inner = nthreads;
    step = (long long)U * nthreads;
    for (long long i = i0; i < n16; i += step) {
      u32x4 v[U];
#pragma unroll
      for (int k = 0; k < U; ++k)
        if (i + k * inner < n16) v[k] = MODE == 1 ? __builtin_nontemporal_load(src + i + k * inner) : src[i + k * inner];
#pragma unroll
      for (int k = 0; k < U; ++k)
        if (i + k * inner < n16) {
          if (MODE) __builtin_nontemporal_store(v[k], dst + i + k * inner);
          else dst[i + k * inner] = v[k];
        }
    }
  }
}

template <int U, int MODE, int CONTIG>
static void run(const char *name, u32x4 *dst, const u32x4 *src, long long bytes, int grid, int threads) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((copyk<U, MODE, CONTIG>), dim3(grid), dim3(threads), 0, 0, dst, src, bytes / 16);
  CK(hipEventRecord(e0, 0));
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((copyk<U, MODE, CONTIG>), dim3(grid), dim3(threads), 0, 0, dst, src, bytes / 16);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("%-34s grid %5d x %4d: %.3f ms  %.0f GB/s\n", name, grid, threads, ms, 2.0 * bytes / ms / 1e6);
}

int main(int argc, char **argv) {
  const long long bytes = (long long)(argc > 1 ? atoi(argv[1]) : 2048) << 20;
  u32x4 *src, *dst; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 2, bytes));
  CK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
  { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 10; ++r) CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-34s %.3f ms  %.0f GB/s\n", "hipMemcpyAsync D2D", ms, 2.0 * bytes / ms / 1e6); }
  for (int threads : {256, 512, 1024})
    for (int grid : {256, 512, 1024, 2048, 8192}) {
      if ((long long)grid * threads > 2 * 1024 * 1024 + 1) continue;
      run<4, 0, 0>("stride U=4 plain", dst, src, bytes, grid, threads);
      run<8, 0, 0>("stride U=8 plain", dst, src, bytes, grid, threads);
      run<16, 0, 0>("stride U=16 plain", dst, src, bytes, grid, threads);
      run<8, 1, 0>("stride U=8 nt load+store", dst, src, bytes, grid, threads);
      run<8, 2, 0>("stride U=8 nt store", dst, src, bytes, grid, threads);
      run<8, 0, 1>("slab U=8 plain", dst, src, bytes, grid, threads);
      run<8, 2, 1>("slab U=8 nt store", dst, src, bytes, grid, threads);
    }
  return 0;
}
