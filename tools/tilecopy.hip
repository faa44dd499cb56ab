// tilecopy -- the memory system's rate for the transposition's access pattern, without the transposition (development tool).
// A workgroup copies a "tile" of R rows x P bytes (R * P = 32 KiB) of an n x n bit matrix to the tile position mirrored at the
// diagonal, where it writes Q-byte pieces (R' = 32 KiB / Q rows): the piece sizes decide how much of a DRAM page and of a
// 128-byte line one request uses.  Tiles are dealt to XCDs in the same super-tile order as gf2_transpose512_kernel (mode 1) or
// in plain order (mode 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// P, Q: bytes per piece on the source / destination side (multiples of 16); one thread moves 16 bytes; 512 threads = 8 KiB per turn
template <int P, int Q>
__global__ __launch_bounds__(512, 8) void tilecopy(u32x4 *__restrict__ D, const u32x4 *__restrict__ S, long long ld16, int tiles_x, int tiles_y, int mode) {
  // source tile: RS = 32768 / P rows x P bytes at (ty * RS, tx * P); destination tile: RD = 32768 / Q rows x Q bytes at (tx * RD', ...) -- for
  // the rate only the shapes matter: the destination tile of (tx, ty) is the (ty, tx)-th tile of shape RD x Q
  constexpr int RS = 32768 / P, RD = 32768 / Q, LP = P / 16, LQ = Q / 16;
  __shared__ u32x4 buf[2048];
  int b = blockIdx.x, tx, ty;
  if (mode) {
    // the 8 XCDs (b mod 8) form a 4 x 2 grid inside a super-tile; each takes a block of bx x by tiles of it (mode >> 8 = bx, by in nibbles; 0: 2 x 4)
    const int bx = (mode >> 8) & 15 ? (mode >> 8) & 15 : 2, by = (mode >> 12) & 15 ? (mode >> 12) & 15 : 4, slots = bx * by;
    const int st = b / (8 * slots), in = b % (8 * slots), sx_n = tiles_x / (4 * bx), sy_n = tiles_y / (2 * by);
    const int x = in & 7, slot = in >> 3;
    int stx = st % sx_n, sty = st / sx_n;
    if (mode & 2) sty = (sty + stx) % sy_n;  // diagonal order: super-tiles in flight together differ in BOTH coordinates
    int px = x & 3, py = x >> 2;
    if (mode & 4) px = (px + stx + sty) & 3, py = (py + stx + (sty >> 2)) & 1;  // an XCD's share of a super-tile rotates from super-tile to super-tile
    tx = stx * 4 * bx + bx * px + slot % bx, ty = sty * 2 * by + by * py + slot / bx;
  } else {
    tx = b % tiles_x, ty = b / tiles_x;
  }
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = tid + 512 * k, r = i / LP, c = i % LP;
    buf[i] = __builtin_nontemporal_load(S + ((long long)ty * RS + r) * ld16 + (long long)tx * LP + c);
  }
  __syncthreads();
  // destination tile: the (tx * tiles_y + ty)-th tile (column-major order of the source tiles: the transposed position when P = Q = 64)
  // of the grid of RD x Q-byte tiles
  const long long didx = (long long)tx * tiles_y + ty;
  const int dtiles_x = (int)(ld16 / LQ);
  const long long dx = didx % dtiles_x, dy = didx / dtiles_x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = tid + 512 * k, r = i / LQ, c = i % LQ;
    __builtin_nontemporal_store(buf[i ^ 1], D + (dy * RD + r) * ld16 + dx * LQ + c);
  }
}

template <int P, int Q>
static void run(u32x4 *D, const u32x4 *S, int n, int mode) {
  const long long ld16 = n / 128;
  const int tiles_x = n / (P * 8), tiles_y = n / (32768 / P);
  if (mode && (tiles_x % 32 || tiles_y % 32 || tiles_x < 32)) return;  // (super-tiles of 8 x 8: a division by tiles_x / 8 = 0 faulted once, profiles/faults/r04_tilecopy_div0)
  if ((long long)tiles_x * P * 8 != n || (long long)tiles_y * (32768 / P) != n || n % 1024) { printf("P=%d: n does not divide\n", P); return; }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((tilecopy<P, Q>), dim3(tiles_x * tiles_y), dim3(512), 0, 0, D, S, ld16, tiles_x, tiles_y, mode);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms / 10 < best) best = ms / 10;
  }
  CK(hipGetLastError());
  const double bytes = 2.0 * n * (double)n / 8;
  printf("P=%4d Q=%4d mode=%d: %.4f ms  %.2f TB/s\n", P, Q, mode, best, bytes / best / 1e9);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 65536;
  const size_t bytes = (size_t)n * n / 8;
  u32x4 *S, *D;
  CK(hipMalloc(&S, bytes)); CK(hipMalloc(&D, bytes));
  CK(hipMemset(S, 1, bytes)); CK(hipMemset(D, 0, bytes));
  for (int mode = 0; mode < 2; ++mode) {
    run<64, 64>(D, S, n, mode);
    run<128, 128>(D, S, n, mode);
    run<256, 256>(D, S, n, mode);
    run<128, 64>(D, S, n, mode);
    run<64, 128>(D, S, n, mode);
    run<512, 512>(D, S, n, mode);
    run<8192, 8192>(D, S, n, mode);
  }
  for (int mode : {3, 5, 7, 1}) {
    run<64, 64>(D, S, n, mode);
    run<128, 128>(D, S, n, mode);
  }
  for (int shape : {0x42, 0x24, 0x44, 0x82, 0x28, 0x81, 0x18, 0x22, 0x11})  // by, bx
    for (int mode : {1, 7}) {
      printf("by=%d bx=%d ", shape >> 4, shape & 15);
      run<64, 64>(D, S, n, mode | ((shape & 15) << 8) | ((shape >> 4) << 12));
    }
  return 0;
}
