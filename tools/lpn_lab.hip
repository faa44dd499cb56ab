// lpn_lab.hip -- development harness for the config-5 kernels (not part of the product).
// Times variants of the single-phase LPN kernels (m4ri-rust_amd/csrc/gf2_lpn.inc) beside the shipped dispatcher
// (gf2k_tallskinny of libm4ri_hip.so) and a plain streaming kernel moving the same bytes, cold (NBUF rotating buffers) and warm,
// and checks every variant bit for bit against a row-by-row reference kernel.
//   usage: lpn_lab [V ...]     (default 64 128 256)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <functional>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

typedef uint64_t u64;
typedef uint32_t u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;
typedef u32 u32x2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const u32x2v lds_cu32x2;
template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl<N>(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}
__device__ __forceinline__ uint4 xor4(uint4 a, uint4 b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }

#define GF2_LPN_LAB 1
#include "../m4ri-rust_amd/csrc/gf2_lpn.inc"

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));      \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

__global__ void fill_kernel(u64 *p, long long nwords, u64 seed) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nwords) return;
  u64 z = seed + 0x9e3779b97f4a7c15ull * (u64)(i + 1);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  p[i] = z ^ (z >> 31);
}

// row-by-row reference: C[i] = XOR of the rows of B selected by the bits of A[i]
__global__ void ref_kernel(const u64 *A, const u64 *B, u64 *C, int m, int wn) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  u64 acc[4] = {0, 0, 0, 0};
  for (int q = 0; q < 4; ++q) {
    u64 x = A[i * 4 + q];
    while (x) {
      const int b = __builtin_ctzll(x);
      x &= x - 1;
      for (int w = 0; w < wn; ++w) acc[w] ^= B[(long long)(64 * q + b) * wn + w];
    }
  }
  for (int w = 0; w < wn; ++w) C[i * wn + w] = acc[w];
}

// plain stream with the traffic of the product: 32 bytes in, 8 wn bytes out per row
template <int WN>
__global__ __launch_bounds__(256) void stream_kernel(const u64 *__restrict__ A, u64 *__restrict__ C, int m) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < m; i += stride) {
    const uint4 *ap = reinterpret_cast<const uint4 *>(A + i * 4);
    const uint4 x = ap[0], y = ap[1];
    if constexpr (WN == 1) {
      C[i] = ((u64)(x.x ^ y.x ^ x.z ^ y.z)) | ((u64)(x.y ^ y.y ^ x.w ^ y.w) << 32);
    } else if constexpr (WN == 2) {
      *reinterpret_cast<uint4 *>(C + i * 2) = xor4(x, y);
    } else {
      uint4 *cp = reinterpret_cast<uint4 *>(C + i * 4);
      cp[0] = make_uint4(x.x ^ 1, x.y, x.z, x.w);
      cp[1] = make_uint4(y.x ^ 1, y.y, y.z, y.w);
    }
  }
}

// contiguous form: a wave instruction covers 1 KiB of consecutive bytes (lane i: 16 bytes at 16 i); output WN words per row.
// NT_HINT: nontemporal loads / stores.  Each thread handles UNR 16-byte pieces per iteration (all requested before any is used).
template <int WN, int UNR, bool NT_HINT>
__global__ __launch_bounds__(256) void stream2_kernel(const u32x4 *__restrict__ A, u32x4 *__restrict__ C, long long npieces) {
  // piece p of A = 16 bytes; output: WN == 4: one 16-byte piece per input piece; WN == 2: one per two; WN == 1: 8 bytes per two
  const long long stride = (long long)gridDim.x * 256 * UNR;
  for (long long base = (long long)blockIdx.x * 256 * UNR + threadIdx.x; base < npieces; base += stride) {
    u32x4 x[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long long p = base + (long long)u * 256;
      if (p < npieces) x[u] = NT_HINT ? __builtin_nontemporal_load(A + p) : A[p];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long long p = base + (long long)u * 256;
      if (p >= npieces) continue;
      if constexpr (WN == 4) {
        u32x4 v = x[u];
        v.x ^= 1;
        if (NT_HINT) __builtin_nontemporal_store(v, C + p);
        else C[p] = v;
      } else if constexpr (WN == 2) {  // half the output: lanes pair up (the even lane stores x ^ its partner's)
        u32x4 v;
        v.x = x[u].x ^ __shfl_xor(x[u].x, 1), v.y = x[u].y ^ __shfl_xor(x[u].y, 1), v.z = x[u].z ^ __shfl_xor(x[u].z, 1),
        v.w = x[u].w ^ __shfl_xor(x[u].w, 1);
        if ((threadIdx.x & 1) == 0) {
          if (NT_HINT) __builtin_nontemporal_store(v, C + (p >> 1));
          else C[p >> 1] = v;
        }
      } else {
        u32 a = x[u].x ^ x[u].z, b = x[u].y ^ x[u].w;
        a ^= __shfl_xor(a, 1), b ^= __shfl_xor(b, 1);
        if ((threadIdx.x & 1) == 0) {
          u64 v = (u64)a | ((u64)b << 32);
          u64 *cp = reinterpret_cast<u64 *>(C) + (p >> 1);
          if (NT_HINT) __builtin_nontemporal_store(v, cp);
          else *cp = v;
        }
      }
    }
  }
}

__global__ void diff_kernel(const u64 *a, const u64 *b, long long nwords, unsigned long long *count) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nwords && a[i] != b[i]) atomicAdd(count, 1ull);
}

typedef hipError_t (*tallskinny_fn)(const u64 *, long long, const u64 *, long long, u64 *, long long, int, int, int, int, hipStream_t);

struct Variant {
  std::string name;
  std::function<void(const u64 *A, const u64 *B, u64 *C, int m, int V)> launch;
};

template <typename K>
static void set_lds(K kernel, int bytes) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
}

static int g_grid_cap = 0;  // LAB_GRID: workgroups per launch at most (persistent batches beyond)
static const char *abl_name(int abl) {
  return abl == 0 ? "" : abl == 1 ? " abl:nobuild" : abl == 2 ? " abl:nolookup" : abl == 3 ? " abl:neither" : abl == 4 ? " stamps" : abl == 8 ? " stamps(build)" : " abl:?";
}

template <int NW, int NT, int RPT, int MODE, int EARLY = 2, int ABL = 0, int STRIDED = 0>
static Variant make_k8() {
  constexpr int ldsb = kLpn8LdsBytes(NW);
  set_lds(gf2_lpn8_kernel<NW, NT, RPT, MODE, EARLY, ABL, STRIDED>, ldsb);
  char nm[96];
  snprintf(nm, sizeof nm, "lpn8<w%d,%d,%d,m%d,e%d%s>%s", NW, NT, RPT, MODE, EARLY, STRIDED ? ",strided" : "", abl_name(ABL));
  return {nm, [](const u64 *A, const u64 *B, u64 *C, int m, int V) {
            unsigned grid = (unsigned)((m + NT * RPT - 1) / (NT * RPT));
            if (g_grid_cap && grid > (unsigned)g_grid_cap) grid = g_grid_cap;
            const int wn = (V + 63) / 64;
            hipLaunchKernelGGL((gf2_lpn8_kernel<NW, NT, RPT, MODE, EARLY, ABL, STRIDED>), dim3(grid), dim3(NT), ldsb, 0, A, 4, B, wn, C, wn, m, 256, V, 0);
          }};
}
template <int NT, int RPT, int MODE, int EARLY = 2, int ABL = 0, int ORD = 0>
static Variant make_k256() {
  set_lds(gf2_lpn256_kernel<NT, RPT, MODE, EARLY, ABL, ORD>, kLpn256LdsBytes);
  char nm[96];
  snprintf(nm, sizeof nm, "lpn256<%d,%d,m%d,e%d,o%d>%s", NT, RPT, MODE, EARLY, ORD, abl_name(ABL));
  return {nm, [](const u64 *A, const u64 *B, u64 *C, int m, int V) {
            unsigned grid = (unsigned)((m + NT * RPT - 1) / (NT * RPT));
            if (g_grid_cap && grid > (unsigned)g_grid_cap) grid = g_grid_cap;
            const int wn = (V + 63) / 64;
            hipLaunchKernelGGL((gf2_lpn256_kernel<NT, RPT, MODE, EARLY, ABL, ORD>), dim3(grid), dim3(NT), kLpn256LdsBytes, 0, A, 4, B, wn, C, wn, m, 256,
                               V, 0);
          }};
}

int main(int argc, char **argv) {
  std::vector<int> Vs;
  for (int i = 1; i < argc; ++i) Vs.push_back(atoi(argv[i]));
  if (Vs.empty()) Vs = {64, 128, 256};
  const int m = getenv("LAB_M") ? atoi(getenv("LAB_M")) : 1 << 20, NBUF = 10;
  g_grid_cap = getenv("LAB_GRID") ? atoi(getenv("LAB_GRID")) : 0;
  const int reps = getenv("LAB_REPS") ? atoi(getenv("LAB_REPS")) : 200;
  tallskinny_fn shipped = nullptr;
  if (void *h = dlopen(getenv("LAB_LIB") ? getenv("LAB_LIB") : "m4ri-rust_amd/lib/libm4ri_hip.so", RTLD_NOW))
    shipped = (tallskinny_fn)dlsym(h, "gf2k_tallskinny");
  if (!shipped) fprintf(stderr, "note: shipped library not found (%s)\n", dlerror());

  std::vector<u64 *> As(NBUF), Cs(NBUF);
  const char *delta_env = getenv("LAB_DELTA");  // placement study: everything in one slab, C_i = A_i + 32 MiB (+ pad) + LAB_DELTA bytes
  if (delta_env) {
    const long long dl = atoll(delta_env);
    const size_t delta = dl < 0 ? 0 : (size_t)dl, slot = (size_t)m * 64 + ((size_t)(getenv("LAB_PAD_MIB") ? atoi(getenv("LAB_PAD_MIB")) : 8) << 20);
    char *slab, *slab2 = nullptr;
    CK(hipMalloc(&slab, slot * NBUF + ((size_t)128 << 20)));
    if (dl < 0) CK(hipMalloc(&slab2, (size_t)m * 32 * NBUF));  // -1: the A buffers end to end in one allocation, the C buffers in another
    for (int i = 0; i < NBUF; ++i) {
      As[i] = reinterpret_cast<u64 *>(dl < 0 ? slab + (size_t)m * 32 * i : slab + slot * i);
      Cs[i] = reinterpret_cast<u64 *>(dl < 0 ? slab2 + (size_t)m * 32 * i : slab + slot * i + (size_t)m * 32 + delta);
      hipLaunchKernelGGL(fill_kernel, dim3((m * 4 + 255) / 256), dim3(256), 0, 0, As[i], (long long)m * 4, 100 + i);
    }
  } else
  for (int i = 0; i < NBUF; ++i) {
    CK(hipMalloc(&As[i], (size_t)m * 32));
    CK(hipMalloc(&Cs[i], (size_t)m * 32 + (size_t)(i + 1) * (1 << 20)));  // different distances between A_i and C_i
    hipLaunchKernelGGL(fill_kernel, dim3((m * 4 + 255) / 256), dim3(256), 0, 0, As[i], (long long)m * 4, 100 + i);
  }
  {  // the stamp buffer of the instrumented variants always exists (a stamped kernel launched without one would write through NULL)
    unsigned long long *st0;
    CK(hipMalloc(&st0, (size_t)4096 * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(lpn_dbg_stamps), &st0, sizeof(st0)));
  }
  u64 *B, *Cref;
  unsigned long long *cnt;
  CK(hipMalloc(&B, 256 * 32));
  CK(hipMalloc(&Cref, (size_t)m * 32));
  CK(hipMalloc(&cnt, 8));
  CK(hipDeviceSynchronize());

  for (int V : Vs) {
    const int wn = (V + 63) / 64;
    // B: 256 rows of wn words, excess bits zero
    std::vector<u64> hb(256 * wn);
    u64 z = 12345 + V;
    for (auto &x : hb) {
      z = z * 6364136223846793005ull + 1442695040888963407ull;
      x = z ^ (z >> 29);
    }
    if (V & 63)
      for (int r = 0; r < 256; ++r) hb[r * wn + wn - 1] &= (1ull << (V & 63)) - 1;
    CK(hipMemcpy(B, hb.data(), hb.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_kernel, dim3((m + 255) / 256), dim3(256), 0, 0, As[0], B, Cref, m, wn);
    CK(hipDeviceSynchronize());

    std::vector<Variant> vs;
    vs.push_back({"stream", [wn](const u64 *A, const u64 *, u64 *C, int m_, int) {
                    if (wn == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(2048), dim3(256), 0, 0, A, C, m_);
                    else if (wn == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(2048), dim3(256), 0, 0, A, C, m_);
                    else hipLaunchKernelGGL(stream_kernel<4>, dim3(2048), dim3(256), 0, 0, A, C, m_);
                  }});
#define STREAM2(UNR, NTH, GRID)                                                                                                  \
  vs.push_back({"stream2<u" #UNR ",nt" #NTH ",g" #GRID ">", [wn](const u64 *A, const u64 *, u64 *C, int m_, int) {                 \
                  const long long np = (long long)m_ * 2;                                                                      \
                  if (wn == 1) hipLaunchKernelGGL((stream2_kernel<1, UNR, NTH>), dim3(GRID), dim3(256), 0, 0, (const u32x4 *)A, (u32x4 *)C, np); \
                  else if (wn == 2) hipLaunchKernelGGL((stream2_kernel<2, UNR, NTH>), dim3(GRID), dim3(256), 0, 0, (const u32x4 *)A, (u32x4 *)C, np); \
                  else hipLaunchKernelGGL((stream2_kernel<4, UNR, NTH>), dim3(GRID), dim3(256), 0, 0, (const u32x4 *)A, (u32x4 *)C, np); \
                }})
    if (getenv("LAB_STREAMS")) {
      STREAM2(1, false, 2048);
      STREAM2(2, false, 2048);
      STREAM2(4, false, 2048);
      STREAM2(4, false, 1024);
      STREAM2(4, false, 512);
      STREAM2(8, false, 1024);
      STREAM2(4, true, 2048);
      STREAM2(4, true, 1024);
      STREAM2(8, true, 1024);
      STREAM2(2, true, 4096);
      STREAM2(1, true, 8192);
    }
    if (shipped)
      vs.push_back({"shipped", [shipped, wn](const u64 *A, const u64 *B_, u64 *C, int m_, int V_) {
                      shipped(A, 4, B_, wn, C, wn, m_, 256, V_, 0, 0);
                    }});
    if (getenv("LAB_SMALL")) {
      if (wn == 1) { vs.push_back(make_k8<1, 512, 4, 2, 3>()); vs.push_back(make_k8<1, 512, 2, 2>()); vs.push_back(make_k8<1, 512, 1, 2, 1>()); vs.push_back(make_k8<1, 256, 2, 2>()); vs.push_back(make_k8<1, 256, 1, 2, 1>()); }
      if (wn == 2) { vs.push_back(make_k8<2, 512, 8, 2, 3>()); vs.push_back(make_k8<2, 512, 4, 2>()); vs.push_back(make_k8<2, 512, 2, 2>()); vs.push_back(make_k8<2, 512, 1, 2, 1>()); vs.push_back(make_k8<2, 1024, 1, 2, 1>()); }
      if (wn >= 3) { vs.push_back(make_k256<512, 8, 2, 3, 0, 1>()); vs.push_back(make_k256<512, 4, 2, 2, 0, 1>()); vs.push_back(make_k256<512, 2, 2, 2, 0, 1>()); vs.push_back(make_k256<512, 1, 2, 1, 0, 1>()); vs.push_back(make_k256<1024, 1, 2, 1, 0, 1>()); }
    } else
    if (wn == 1) {
      vs.push_back(make_k8<1, 512, 4, 2>());
      vs.push_back(make_k8<1, 512, 4, 2, 3>());
      vs.push_back(make_k8<1, 512, 4, 2, 3, 0, 1>());
      vs.push_back(make_k8<1, 512, 8, 2>());
      vs.push_back(make_k8<1, 512, 8, 2, 3>());
      vs.push_back(make_k8<1, 512, 8, 2, 4>());
      vs.push_back(make_k8<1, 256, 8, 2, 3>());
      vs.push_back(make_k8<1, 256, 16, 2, 3>());
      vs.push_back(make_k8<1, 1024, 2, 2>());
    }
    if (wn == 2 && !getenv("LAB_SMALL")) {
      vs.push_back(make_k8<2, 1024, 4, 2, 4>());
      vs.push_back(make_k8<2, 512, 8, 2>());
      vs.push_back(make_k8<2, 512, 8, 2, 3>());
      vs.push_back(make_k8<2, 512, 8, 2, 3, 0, 1>());
      vs.push_back(make_k8<2, 512, 8, 2, 4>());
      vs.push_back(make_k8<2, 512, 8, 2, 6>());
      vs.push_back(make_k8<2, 512, 4, 2>());
      vs.push_back(make_k8<2, 512, 4, 2, 3>());
      vs.push_back(make_k8<2, 256, 16, 2, 3>());
      vs.push_back(make_k8<2, 512, 8, 2, 2, 3>());
    }
    if (wn >= 3 && !getenv("LAB_SMALL")) {
      vs.push_back(make_k256<1024, 4, 2, 2, 0, 1>());
      vs.push_back(make_k256<1024, 4, 2, 2, 0, 5>());
      vs.push_back(make_k256<1024, 4, 2, 3, 0, 1>());
      vs.push_back(make_k256<512, 8, 2, 3, 0, 1>());
      vs.push_back(make_k256<512, 8, 2, 3, 0, 5>());
      vs.push_back(make_k256<512, 8, 2, 4, 0, 1>());
      vs.push_back(make_k256<512, 8, 2, 2, 3>());
      vs.push_back(make_k256<512, 8, 2, 3, 0, 13>());
      vs.push_back(make_k256<1024, 4, 2, 2, 0, 13>());
      vs.push_back(make_k256<512, 8, 2, 3, 3, 13>());
      vs.push_back(make_k256<512, 8, 2, 3, 1, 5>());
      vs.push_back(make_k256<512, 8, 2, 3, 2, 5>());
      vs.push_back(make_k256<512, 8, 2, 3, 3, 5>());
      vs.push_back(make_k256<1024, 4, 2, 2, 4, 1>());
      vs.push_back(make_k256<512, 8, 2, 3, 4, 1>());
      vs.push_back(make_k256<512, 8, 2, 3, 4, 5>());
      vs.push_back(make_k256<512, 8, 2, 3, 8, 1>());
      vs.push_back(make_k256<512, 8, 2, 3, 4, 13>());
      vs.push_back(make_k256<512, 8, 2, 3, 8, 13>());
    }
    const char *only = getenv("LAB_ONLY");      // run only the variants whose name contains this
    const bool profile = getenv("LAB_PROFILE");  // a few launches per variant, no timing loops (counter runs)
    for (auto &v : vs) {
      if (only && v.name.find(only) == std::string::npos) continue;
      if (profile && v.name.find("stamps") != std::string::npos) continue;  // (a counter run wants the plain kernels)
      if (profile) {
        for (int i = 0; i < 8; ++i) v.launch(As[i % NBUF], B, Cs[(3 * i + 1) % NBUF], m, V);
        CK(hipDeviceSynchronize());
        continue;
      }
      // correctness on buffer 0
      bool ok = true;
      if (v.name.find("stream") == std::string::npos && v.name.find("stamps") == std::string::npos && v.name.find("abl:") == std::string::npos) {
        CK(hipMemset(Cs[0], 0xa5, (size_t)m * wn * 8));
        v.launch(As[0], B, Cs[0], m, V);
        CK(hipGetLastError());
        CK(hipMemset(cnt, 0, 8));
        hipLaunchKernelGGL(diff_kernel, dim3((unsigned)(((long long)m * wn + 255) / 256)), dim3(256), 0, 0, Cs[0], Cref, (long long)m * wn, cnt);
        unsigned long long bad = 0;
        CK(hipMemcpy(&bad, cnt, 8, hipMemcpyDeviceToHost));
        ok = bad == 0;
        if (!ok) printf("V=%d %-16s MISMATCH in %llu words\n", V, v.name.c_str(), bad);
      }
      if (v.name.find("stamps") != std::string::npos) {  // per-wave timeline of ONE cold launch (after a few others)
        const int nw_ = 4096;
        unsigned long long *st;
        CK(hipMalloc(&st, (size_t)nw_ * 16 * 8));
        CK(hipMemset(st, 0, (size_t)nw_ * 16 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(lpn_dbg_stamps), &st, sizeof(st)));
        for (int i = 0; i < 30; ++i) v.launch(As[i % NBUF], B, Cs[(3 * i + 1) % NBUF], m, V);
        CK(hipDeviceSynchronize());
        CK(hipMemset(st, 0, (size_t)nw_ * 16 * 8));
        CK(hipDeviceSynchronize());
        v.launch(As[0], B, Cs[1], m, V);  // (A_0 was last read 20 launches = 640 MiB of traffic ago: cold)
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)nw_ * 16);
        CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        // s_memrealtime (100 MHz) is one counter for the chip; s_memtime counts shader cycles PER XCD (the XCDs' counters are
        // unrelated): differences are taken inside a wave only, the clock is the median over waves of d(memtime) / d(memrealtime)
        unsigned long long t0r = ~0ull, t1r = 0;
        std::vector<double> clk;
        for (int w = 0; w < nw_; ++w) {
          if (!h[w * 16]) continue;
          t0r = std::min(t0r, h[w * 16]);
          t1r = std::max(t1r, h[w * 16 + 14]);
          if (h[w * 16 + 14] > h[w * 16]) clk.push_back((double)(h[w * 16 + 13] - h[w * 16 + 1]) / ((double)(h[w * 16 + 14] - h[w * 16]) * 10.0));
        }
        std::sort(clk.begin(), clk.end());
        const double ghz = clk.empty() ? 2.0 : clk[clk.size() / 2];
        printf("  stamps of one cold launch: kernel span %.2f us (realtime); clock held inside the waves: median %.3f GHz (p10 %.3f, p90 %.3f)\n",
               (t1r - t0r) / 100.0, ghz, clk.empty() ? 0.0 : clk[clk.size() / 10], clk.empty() ? 0.0 : clk[clk.size() * 9 / 10]);
        const char *names[16] = {"", "start", "B staged", "built", "tables ready", "row0", "row1", "row2", "row3", "row4", "row5", "row6", "row7", "end"};
        auto stat = [&](const char *what, std::vector<double> &xs) {
          if (xs.empty()) return;
          std::sort(xs.begin(), xs.end());
          printf("  %-28s min %7.0f  median %7.0f  p90 %7.0f  max %7.0f cycles  (median %.2f us)\n", what, xs[0], xs[xs.size() / 2], xs[xs.size() * 9 / 10],
                 xs.back(), xs[xs.size() / 2] / ghz / 1e3);
        };
        {
          std::vector<double> xs;  // (realtime ticks of 10 ns, shown as cycles at the median clock)
          for (int w = 0; w < nw_; ++w)
            if (h[w * 16]) xs.push_back((double)(h[w * 16] - t0r) * 10.0 * ghz);
          stat("start after first start", xs);
        }

        int prev = 1;
        for (int k = 2; k <= 13; ++k) {
          std::vector<double> xs;
          for (int w = 0; w < nw_; ++w)
            if (h[w * 16 + k] && h[w * 16 + prev]) xs.push_back((double)(h[w * 16 + k] - h[w * 16 + prev]));
          if (xs.empty()) continue;
          char nm[64];
          snprintf(nm, sizeof nm, "%s -> %s", names[prev], names[k]);
          stat(nm, xs);
          prev = k;
        }
        {
          std::vector<double> xs;
          for (int w = 0; w < nw_; ++w)
            if (h[w * 16]) xs.push_back((double)(h[w * 16 + 13] - h[w * 16 + 1]));
          stat("start -> end (wave lifetime)", xs);
        }
        CK(hipFree(st));
        continue;
      }
      double res[2];
      for (int mode = 0; mode < 2; ++mode) {  // 0 warm, 1 cold
        const int k = mode ? NBUF : 1;
        for (int i = 0; i < 400; ++i) v.launch(As[i % k], B, Cs[(3 * i + 1) % k], m, V);  // clocks up
        CK(hipDeviceSynchronize());
        std::vector<double> ts;
        for (int run = 0; run < 5; ++run) {
          auto t0 = std::chrono::steady_clock::now();
          for (int i = 0; i < reps; ++i) v.launch(As[i % k], B, Cs[(3 * i + 1) % k], m, V);
          CK(hipDeviceSynchronize());
          ts.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps);
        }
        std::sort(ts.begin(), ts.end());
        res[mode] = ts[2];
      }
      const double bytes = (double)m * 32 + 256.0 * wn * 8 + (double)m * wn * 8;
      printf("V=%3d %-16s %s warm %6.2f us  cold %6.2f us  (%.0f GB/s = %.3f of 8 TB/s)\n", V, v.name.c_str(), ok ? "ok " : "BAD", res[0] * 1e6,
             res[1] * 1e6, bytes / res[1] / 1e9, bytes / res[1] / 8e12);
      fflush(stdout);
    }
  }
  return 0;
}
