// kbench -- kernel micro-benchmark for the M4RM tile kernel variants (development tool, not shipped).
//   kbench <n> <batch> <reps> <cfg> [<cfg> ...]          (n may be m,l,n)
// env: APACK=1 (row-group-packed A for the variants that read it), KSPLIT=k (uniform split-K), NREM=r NSEG=s (stream-K split of the
// v8 family, cfg 9-12: the last r tiles cut into s segments; NREM=-1: all tiles; NREM=-2: the tiles of the last incomplete round)
// Runs each variant on `batch` independent n x n x n products of seeded random bits, checks every variant's
// output bit for bit against cfg 0, prints avg kernel time (HIP events) and derived rates.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <utility>
#include "../m4ri-rust_amd/csrc/gf2_kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "usage: kbench n batch reps cfg...\n"); return 2; }
  int n = atoi(argv[1]);
  const int batch = atoi(argv[2]), reps = atoi(argv[3]);
  const long long ld = n / 64, words = (long long)n * ld;
  uint64_t *A, *B, *C, *Cref;
  CK(hipMalloc(&A, words * 8 * batch)); CK(hipMalloc(&B, words * 8 * batch));
  CK(hipMalloc(&C, words * 8 * batch)); CK(hipMalloc(&Cref, words * 8 * batch));
  CK(gf2k_fill_random(A, ld, n * batch, n, 1, 0, 0, 0, 0));
  CK(gf2k_fill_random(B, ld, n * batch, n, 2, 0, 0, 0, 0));
  int *diff; CK(hipMalloc(&diff, 4));
  gf2k_mul_args a{};
  a.A = A; a.B = B; a.lda = a.ldb = a.ldc = ld; a.sA = a.sB = a.sC = words; a.m = a.l = a.n = n; a.batch = batch; a.ksplit = getenv("KSPLIT") ? atoi(getenv("KSPLIT")) : 1;
  const int nrem_env = getenv("NREM") ? atoi(getenv("NREM")) : 0, nseg_env = getenv("NSEG") ? atoi(getenv("NSEG")) : 0;
  uint64_t *Pws = nullptr;
  const long long pws_words = 2ll * 4096 * 4096 * 8;  // room for 4096 segments of the tallest tile (1 GiB)
  if (nrem_env || a.ksplit > 1) CK(hipMalloc(&Pws, pws_words * 8));
  // chunk-packed copy of B for the BPACK kernel variants
  const int nc = gf2k_packB_chunks(n), tiles_n = (n + 2047) / 2048;
  const long long bpBlocks = (long long)tiles_n * nc;
  uint32_t *Bp; CK(hipMalloc(&Bp, (size_t)bpBlocks * 2048 * batch));
  CK(gf2k_packB(Bp, bpBlocks * 512, B, ld, words, n, n, batch, 0));
  a.Bp = Bp; a.sBp = bpBlocks; a.bp_nc = nc;
  // APACK=1: A row-group packed (gf2k_packA), as the Strassen split / the plain-product prologue hand it to v6 / v7
  uint64_t *Apk = nullptr;
  if (getenv("APACK") && atoi(getenv("APACK"))) {
    CK(hipMalloc(&Apk, words * 8 * batch));
    for (int b = 0; b < batch; ++b) CK(gf2k_packA(Apk + b * words, ld, A + b * words, ld, n, (int)ld, 0));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  a.C = Cref;
  CK(gf2k_m4rm(a, 0, 0)); CK(hipDeviceSynchronize());
  for (int i = 4; i < argc; ++i) {
    const int cfg = atoi(argv[i]);
    a.C = C;
    const bool pk = Apk && (cfg == 8 || (cfg >= 9 && cfg <= 24) || (cfg >= 90 && cfg < 100));
    a.P = nullptr; a.p_words = 0; a.n_rem = a.nseg = 0;
    if (cfg >= 9 && cfg <= 24 && Pws) {
      const int tcols = gf2k_m4rm_cols_per_tile(cfg);
      const long long T = (long long)((n + gf2k_m4rm_rows_per_tile(cfg) - 1) / gf2k_m4rm_rows_per_tile(cfg)) * ((n + tcols - 1) / tcols) * batch;
      a.P = Pws; a.p_words = pws_words;
      a.n_rem = nrem_env == -1 ? (int)T : nrem_env == -2 ? (int)(T < 256 ? T : T % 256) : nrem_env;
      a.nseg = nseg_env;
    }
    if (cfg == 96) CK(hipMemset(Bp, 0, (size_t)bpBlocks * 2048 * batch));
    a.A = pk ? Apk : A;
    a.a_packed = pk ? 1 : 0;
    CK(hipMemset(C, 0xff, words * 8 * batch));
    CK(gf2k_m4rm(a, cfg, 0)); CK(hipDeviceSynchronize());
    CK(hipMemset(diff, 0, 4));
    CK(gf2k_diff(C, ld, Cref, ld, n * batch, n, diff, 0));
    int h = -1; CK(hipMemcpy(&h, diff, 4, hipMemcpyDeviceToHost));
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) CK(gf2k_m4rm(a, cfg, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double macs = (double)n * n * n * batch;
    printf("cfg %d  n=%d batch=%d  %s  %.3f ms  %.2f Tbitmac/s  lds-read %.1f TB/s\n", cfg, n, batch,
           h == 0 ? "OK " : "MISMATCH", ms, macs / ms / 1e9, macs / 64 / ms / 1e9);
    if (cfg == 96) {  // phase stamps of the v7 kernel (five u64 per workgroup in Bp): per-tile overheads
      const long long nwg = (long long)((n + 4095) / 4096) * ((n + 511) / 512) * batch;
      std::vector<unsigned long long> d(nwg * 5);
      CK(hipMemcpy(d.data(), Bp, d.size() * 8, hipMemcpyDeviceToHost));
      double pro = 0, loop = 0, epi = 0;
      std::vector<std::pair<unsigned long long, std::pair<unsigned long long, unsigned long long>>> byc;  // (cu key, (start, end))
      for (long long w = 0; w < nwg; ++w) {
        pro += (double)(d[5 * w + 1] - d[5 * w]); loop += (double)(d[5 * w + 2] - d[5 * w + 1]); epi += (double)(d[5 * w + 3] - d[5 * w + 2]);
        const unsigned hw = (unsigned)d[5 * w + 4]; const unsigned xcc = (unsigned)(d[5 * w + 4] >> 32) & 15;
        const unsigned long long key = ((unsigned long long)xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
        byc.push_back({key, {d[5 * w], d[5 * w + 3]}});
      }
      std::sort(byc.begin(), byc.end());
      double gap = 0; long long ngap = 0; int ncu = 0;
      for (size_t i = 0; i < byc.size(); ++i) {
        if (i == 0 || byc[i].first != byc[i - 1].first) { ++ncu; continue; }
        gap += (double)((long long)byc[i].second.first - (long long)byc[i - 1].second.second); ++ngap;
      }
      printf("   per workgroup (10 ns ticks -> us): prologue %.2f  loop %.2f  epilogue+stores %.2f; %d CUs, gap between consecutive workgroups on a CU %.2f us\n",
             pro / nwg / 100, loop / nwg / 100, epi / nwg / 100, ncu, ngap ? gap / ngap / 100 : 0.0);
    }
    if (cfg == 49) {
      unsigned long long d[8]; CK(gf2k_dbg_sec(d));
      for (int w = 0; w < 2; ++w)
        printf("   wave %s: per chunk: steps %.0f  loads %.0f  barrier %.0f cycles (%llu chunks)\n", w ? "last" : "0", (double)d[4*w]/d[4*w+3], (double)d[4*w+1]/d[4*w+3], (double)d[4*w+2]/d[4*w+3], d[4*w+3]);
    }
    fflush(stdout);
  }
  return 0;
}
