// overlap_probe -- can an HBM-streaming pass run UNDER the tile kernel?  (development tool)
//   overlap_probe [leaves=343] [copy_MiB=1350] [grids...]
// The leaf launch of the 65536^3 product (343 packed leaves of 4096^3 = one top-level Strassen product, ~3.5 ms, LDS-bound, one
// workgroup per CU with all of its registers) on one stream; a grid-stride copy (read + write, the shape of a split / merge pass)
// on a second stream with a grid of `grid` workgroups of 1024 threads.  Prints each alone and both together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../m4ri-rust_amd/csrc/gf2_kernels.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void stream_copy(uint4 *__restrict__ dst, const uint4 *__restrict__ src, long long n16) {
  const long long stride = (long long)gridDim.x * 1024;
  for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n16; i += 4 * stride) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i + k * stride < n16 ? src[i + k * stride] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k * stride < n16) dst[i + k * stride] = v[k];
  }
}

static float ms_between(hipEvent_t a, hipEvent_t b) { float t; CK(hipEventElapsedTime(&t, a, b)); return t; }

int main(int argc, char **argv) {
  const int leaves = argc > 1 ? atoi(argv[1]) : 343;
  const long long copy_bytes = (long long)(argc > 2 ? atoi(argv[2]) : 1350) << 20;
  std::vector<int> grids;
  for (int i = 3; i < argc; ++i) grids.push_back(atoi(argv[i]));
  if (grids.empty()) grids = {32, 64, 128, 256, 1024};
  const int n = 4096;
  const long long ld = n / 64, words = (long long)n * ld;
  uint64_t *A, *Ap, *B, *C;
  CK(hipMalloc(&A, words * 8 * leaves)); CK(hipMalloc(&Ap, words * 8 * leaves));
  CK(hipMalloc(&B, words * 8 * leaves)); CK(hipMalloc(&C, words * 8 * leaves));
  CK(gf2k_fill_random(A, ld, n * leaves, n, 1, 0, 0, 0, 0));
  CK(gf2k_fill_random(B, ld, n * leaves, n, 2, 0, 0, 0, 0));
  for (int b = 0; b < leaves; ++b) CK(gf2k_packA(Ap + b * words, ld, A + b * words, ld, n, (int)ld, 0));
  uint4 *src, *dst;
  CK(hipMalloc(&src, copy_bytes)); CK(hipMalloc(&dst, copy_bytes));
  CK(hipMemset(src, 1, copy_bytes));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  gf2k_mul_args a{};
  a.A = Ap; a.B = B; a.C = C; a.lda = a.ldb = a.ldc = ld; a.sA = a.sB = a.sC = words; a.m = a.l = a.n = n; a.batch = leaves; a.a_packed = 1;
  hipEvent_t e0, e1, f0, f1, g0;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1)); CK(hipEventCreate(&g0));
  auto leaf = [&](hipStream_t s) { CK(gf2k_m4rm(a, 9, s)); };
  auto copy = [&](hipStream_t s, int grid) { hipLaunchKernelGGL(stream_copy, dim3(grid), dim3(1024), 0, s, dst, src, copy_bytes / 16); };
  for (int w = 0; w < 10; ++w) leaf(s1);  // clocks up
  CK(hipDeviceSynchronize());
  float t_leaf = 0;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, s1)); leaf(s1); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
    t_leaf += ms_between(e0, e1) / 5;
  }
  printf("leaf launch alone (%d leaves): %.3f ms\n", leaves, t_leaf);
  for (int grid : grids) {
    float t_copy = 0, t_both = 0, t_leaf_in = 0, t_copy_in = 0;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(f0, s2)); copy(s2, grid); CK(hipEventRecord(f1, s2)); CK(hipEventSynchronize(f1));
      t_copy += ms_between(f0, f1) / 5;
    }
    for (int r = 0; r < 5; ++r) {
      CK(hipDeviceSynchronize());
      // the leaf launch first, the copy behind a short head start of the leaf kernel (it then finds every CU taken)
      CK(hipEventRecord(e0, s1)); leaf(s1); CK(hipEventRecord(e1, s1));
      CK(hipEventRecord(f0, s2)); copy(s2, grid); CK(hipEventRecord(f1, s2));
      CK(hipEventSynchronize(e1)); CK(hipEventSynchronize(f1));
      t_leaf_in += ms_between(e0, e1) / 5;
      t_copy_in += ms_between(f0, f1) / 5;
      const float end = std::max(ms_between(e0, e1), ms_between(e0, f1));
      t_both += end / 5;
    }
    printf("copy grid %5d: alone %.3f ms (%.0f GB/s); together: leaf %.3f ms, copy %.3f ms, both done after %.3f ms (sum alone %.3f)\n", grid,
           t_copy, 2.0 * copy_bytes / t_copy / 1e6, t_leaf_in, t_copy_in, t_both, t_leaf + t_copy);
  }
  return 0;
}
