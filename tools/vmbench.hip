// vmbench.hip -- per-CU throughput of vector-memory load instructions whose data sits in L1/L2 (development tool).
//   hipcc --offload-arch=gfx950 -O3 -o vmbench vmbench.hip && ./vmbench
// One workgroup of W waves on one CU issues N back-to-back buffer loads of a given shape; reports cycles per instruction
// for the CU.  Shapes: b32 with 4 distinct dwords per wave (16-lane groups share an address: how the tile kernel reads
// A), b32 coalesced (256 B per wave: how it reads a row of B), b64 / b128 coalesced.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k(const u32 *buf, unsigned long long *out, u32 *sink, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)buf, (short)0, 1 << 20, 0x00020000);
  u32 voff;
  if (MODE == 0) voff = (u32)(lane >> 4) * 512u + wave * 2048u;      // 4 distinct dwords, 512 B apart
  else if (MODE == 1) voff = (u32)lane * 4u + wave * 2048u;           // 256 B coalesced
  else if (MODE == 2) voff = (u32)lane * 8u + wave * 2048u;           // 512 B
  else voff = (u32)lane * 16u + wave * 2048u;                         // 1 KiB
  u32 acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int so = ((it * 8 + u) & 15) * 32768 % (1 << 19);
      if (MODE <= 1) acc ^= __builtin_amdgcn_raw_buffer_load_b32(rs, voff, so, 0);
      else if (MODE == 2) { u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, so, 0); acc ^= x.x ^ x.y; }
      else { u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so, 0); acc ^= x.x ^ x.y ^ x.z ^ x.w; }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
static void run(const char *name, int waves, const u32 *buf, unsigned long long *out, u32 *sink) {
  const int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(waves * 64), 0, 0, buf, out, sink, 10);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(waves * 64), 0, 0, buf, out, sink, iters);
  CK(hipDeviceSynchronize());
  unsigned long long c;
  CK(hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost));
  const double per = (double)c / (iters * 8.0 * waves);
  printf("%-44s waves=%2d  cycles per load instruction (CU-wide) = %6.2f\n", name, waves, per);
}

int main() {
  u32 *buf, *sink;
  unsigned long long *out;
  CK(hipMalloc(&buf, 1 << 20));
  CK(hipMemset(buf, 1, 1 << 20));
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&sink, 64));
  for (int w : {1, 4, 8, 16}) {
    run<0>("b32, 4 distinct dwords per wave (A pattern)", w, buf, out, sink);
    run<1>("b32 coalesced, 256 B per wave (B-row pattern)", w, buf, out, sink);
    run<2>("b64 coalesced, 512 B per wave", w, buf, out, sink);
    run<3>("b128 coalesced, 1 KiB per wave", w, buf, out, sink);
  }
  return 0;
}
