// graphbench.hip -- cost of a chain of dependent small kernels: stream launches against one hipGraph launch (development tool).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void tiny(unsigned long long *p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
int main() {
  unsigned long long *d;
  CK(hipMalloc(&d, 64));
  CK(hipMemset(d, 0, 64));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const int N = 256, REP = 20;
  for (int grid : {1, 256}) {
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d);
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REP; ++r)
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d);
    CK(hipStreamSynchronize(s));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (REP * N);
    printf("grid %3d: stream launches  %.2f us per dependent kernel\n", grid, us);
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (REP * N);
    printf("grid %3d: one graph launch %.2f us per dependent kernel\n", grid, us);
  }
  return 0;
}
