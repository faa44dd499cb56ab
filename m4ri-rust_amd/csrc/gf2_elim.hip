// gf2_elim.hip -- row echelon forms over GF(2) on gfx950, device resident.
//
// Replaces the elimination entry points the friendly layer reaches (paths relative to /root/reference):
//   mzd_echelonize / _m4ri / _pluq   m4ri-sys/src/echelonform.rs:16-37   (BinMatrix::echelonize / rank,
//                                                                          binary_matrix.rs:246-261)
//   mzd_inv_m4ri                     m4ri-sys/src/brilliantrussian.rs:201-208  (BinMatrix::inverted, :263-268)
//   mzd_solve_left                   m4ri-sys/src/solve.rs:12-29          (solve_left, binary_matrix.rs:575-586)
//
// Blocked Gauss-Jordan.  Columns are processed in blocks of up to 2048 (32 words).  Inside a block the columns of one
// 64-bit word form a step:
//   pivot   one workgroup scans the word of every active row (rows >= rank), keeps a GF(2) basis of at most 64 words
//           in LDS (1024 candidates reduced against it per pass, insertion by ballot inside one wave), then
//           back-substitutes so that the chosen rows are reduced on the pivot columns, sorts them by column and emits
//           the row moves that bring them to rows [rank, rank+np);
//   ptab    the np reduced pivot rows, restricted to the block's columns plus the block's tracking matrix U;
//   update  every other row XORs the pivot rows its word selects (table in LDS, one wave per row, lane = word);
//   gather / scatter   the <= 128 row moves (whole rows, so the columns right of the block travel with them).
// U (rows x 2048 bits) records, for each row, which of the block's pivot rows (as they were when the block started)
// have been added to it; when the block is finished, everything right of it is updated with ONE product
//   A[:, right] ^= U' * A[pivot rows of the block, right]
// through the M4RM tile kernel (gf2_kernels.hip), which is where almost all bit operations of a large elimination go.
// All per-step decisions live in a device-side state record, so a block is enqueued without host round trips.
#include "gf2_kernels.h"

typedef uint64_t u64;

__device__ __forceinline__ u64 shfl64(u64 v, int src) {
  const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src), hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 readfirst64(u64 v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((u64)hi << 32) | lo;
}

// ---------------------------------------------------------------------------------------------
// pivot search in one word column
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gf2_elim_pivot_kernel(const u64 *__restrict__ A, long long lda, int m, long long wc,
                                                              u64 colmask, gf2k_elim_state *st, int *pivcols) {
  __shared__ u64 b_word[64], b_trk[64];
  __shared__ int b_row[64], b_col[64];
  __shared__ int s_nb;
  __shared__ int s_nz[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r_cur = st->r_cur;
  if (tid == 0) s_nb = 0;
  __syncthreads();
  for (int base = r_cur; base < m; base += 1024) {
    const int nb0 = s_nb;
    if (nb0 == 64) break;
    const int i = base + tid;
    u64 w = i < m ? (A[(long long)i * lda + wc] & colmask) : 0, t = 0;
    // w: candidate reduced against the basis; t: which basis rows (as originally read) were added to it
    for (int k = 0; k < nb0; ++k)
      if ((w >> b_col[k]) & 1) {
        w ^= b_word[k];
        t ^= b_trk[k];
      }
    const u64 nzb = __ballot(w != 0);
    if (lane == 0) s_nz[wave] = nzb != 0;
    __syncthreads();
    for (int wv = 0; wv < 16; ++wv) {
      if (!s_nz[wv]) continue;  // uniform over the workgroup
      if (wave == wv) {
        int nbl = s_nb;
        for (int k = nb0; k < nbl; ++k)  // vectors inserted by earlier waves of this pass
          if ((w >> b_col[k]) & 1) {
            w ^= b_word[k];
            t ^= b_trk[k];
          }
        u64 mask = __ballot(w != 0);
        while (mask && nbl < 64) {
          const int p = __builtin_ctzll(mask);  // lowest row first
          const u64 pw = shfl64(w, p);
          const u64 pt = shfl64(t, p) | (1ull << nbl);
          const int c = __builtin_ctzll(pw);
          if (lane == 0) {
            b_word[nbl] = pw;
            b_trk[nbl] = pt;
            b_row[nbl] = base + wv * 64 + p;
            b_col[nbl] = c;
          }
          if (lane == p) {
            w = 0;
          } else if ((w >> c) & 1) {
            w ^= pw;
            t ^= pt;
          }
          ++nbl;
          mask = __ballot(w != 0);
        }
        if (lane == 0) s_nb = nbl;
      }
      __syncthreads();
    }
    __syncthreads();  // s_nz is rewritten by the next pass
  }
  __syncthreads();
  if (wave != 0) return;

  const int np = s_nb;
  u64 bw = lane < np ? b_word[lane] : 0, bt = lane < np ? b_trk[lane] : 0;
  const int c = lane < np ? b_col[lane] : 0, row = lane < np ? b_row[lane] : -1;
  // vector k is already clear on the pivot columns of earlier vectors; clear the later ones (Gauss-Jordan)
  for (int kk = np - 1; kk >= 1; --kk) {
    const u64 pw = shfl64(bw, kk), pt = shfl64(bt, kk);
    const int pc = __shfl(c, kk);
    if (lane < kk && ((bw >> pc) & 1)) {
      bw ^= pw;
      bt ^= pt;
    }
  }
  u64 pcmask = lane < np ? (1ull << c) : 0;
  for (int o = 32; o; o >>= 1) pcmask |= shfl64(pcmask, lane ^ o);
  const int pos = lane < np ? __popcll(pcmask & ((1ull << c) - 1)) : 64;  // order by pivot column
  u64 nt = 0;
  for (int k = 0; k < np; ++k) {
    const int pk = __shfl(pos, k);
    if ((bt >> k) & 1) nt |= 1ull << pk;
  }
  if (lane < np) {
    st->piv_row[pos] = row;
    st->piv_col[pos] = c;
    st->trk[pos] = nt;
    pivcols[r_cur + pos] = (int)(wc * 64 + c);
    st->mv_src[pos] = row;
    st->mv_dst[pos] = r_cur + pos;
    st->mv_piv[pos] = pos;
  }
  // rows inside [r_cur, r_cur+np) that are not pivots trade places with the pivot rows coming from below
  bool is_src = false;
  for (int k = 0; k < np; ++k) is_src |= (__shfl(row, k) == r_cur + lane);
  const bool displaced = lane < np && !is_src, vacated = lane < np && row >= r_cur + np;
  const u64 dmask = __ballot(displaced), vmask = __ballot(vacated);
  const int q = __popcll(dmask & ((1ull << lane) - 1));
  u64 vm = vmask;
  for (int i = 0; i < q && vm; ++i) vm &= vm - 1;
  const int vk = vm ? __builtin_ctzll(vm) : 0;
  const int vrow = __shfl(row, vk);
  if (displaced) {
    st->mv_src[np + q] = r_cur + lane;
    st->mv_dst[np + q] = vrow;
    st->mv_piv[np + q] = -1;
  }
  if (lane == 0) {
    st->np = np;
    st->nmoves = np + __popcll(dmask);
    st->pcmask = pcmask;
    st->jbase = r_cur - st->r0;
    st->r_cur = r_cur + np;
  }
}

// reduced pivot rows over the block's columns [c0w, c0w+sw) and the tracking words [0, uw): one wave per pivot
__global__ __launch_bounds__(64) void gf2_elim_ptab_kernel(const u64 *__restrict__ A, long long lda, long long c0w, int sw,
                                                           const u64 *__restrict__ U, long long ldu, int uw,
                                                           const gf2k_elim_state *st, u64 *__restrict__ ptab) {
  const int k = blockIdx.x, lane = threadIdx.x;
  if (k >= st->np) return;
  const int jbase = st->jbase;
  if (lane >= sw + uw) return;
  u64 acc = 0;
  for (u64 t = st->trk[k]; t; t &= t - 1) {
    const int k2 = __builtin_ctzll(t);
    const long long r = st->piv_row[k2];
    if (lane < sw) {
      acc ^= A[r * lda + c0w + lane];
    } else {
      const int u = lane - sw, j = jbase + k2;  // row r becomes pivot j of the block: its own unit bit
      acc ^= U[r * ldu + u] ^ ((j >> 6) == u ? 1ull << (j & 63) : 0);
    }
  }
  ptab[k * 64 + lane] = acc;
}

// every row adds the pivot rows selected by its bits on the pivot columns (pivot rows themselves are rewritten by the
// scatter afterwards, whatever lands in them here is discarded)
__global__ __launch_bounds__(256) void gf2_elim_update_kernel(u64 *__restrict__ A, long long lda, int m, int full,
                                                              long long c0w, int sw, int j, u64 *__restrict__ U,
                                                              long long ldu, int uw, const gf2k_elim_state *st,
                                                              const u64 *__restrict__ ptab) {
  __shared__ u64 tab[64 * 64];
  const int np = st->np;
  if (np == 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < np * 64; i += 256) tab[i] = ptab[i];
  __syncthreads();
  const u64 pcmask = st->pcmask;
  const int rows_lo = full ? 0 : st->r0;
  const long long wc = c0w + j;
  const int nS = sw - j;
  const bool isS = lane < nS, act = lane < nS + uw;
  const int tword = isS ? j + lane : sw + (lane - nS);
  u64 *const base = isS ? A + wc + lane : U + (lane - nS);
  const long long ld = isS ? lda : ldu;
  const int gw = blockIdx.x * 4 + (tid >> 6), nw = gridDim.x * 4;
  constexpr int RG = 4;
  for (long long r0 = rows_lo + (long long)gw * RG; r0 < m; r0 += (long long)nw * RG) {
    u64 sel[RG], old[RG];
#pragma unroll
    for (int q = 0; q < RG; ++q) {
      const long long r = r0 + q;
      sel[q] = r < m ? A[r * lda + wc] & pcmask : 0;
      old[q] = (act && r < m) ? base[r * ld] : 0;
    }
#pragma unroll
    for (int q = 0; q < RG; ++q) {
      u64 s = readfirst64(sel[q]), acc = 0;
      if (!s) continue;
      for (; s; s &= s - 1) {
        const int b = __builtin_ctzll(s);
        const int pos = __popcll(pcmask & ((1ull << b) - 1));
        acc ^= tab[pos * 64 + (act ? tword : 0)];
      }
      if (act && r0 + q < m) base[(r0 + q) * ld] = old[q] ^ acc;
    }
  }
}

// row moves, two phases through a scratch buffer: pivot rows take their block columns and tracking words from ptab
__global__ __launch_bounds__(256) void gf2_elim_gather_kernel(const u64 *__restrict__ A, long long lda, long long aw,
                                                              long long c0w, int sw, const u64 *__restrict__ U,
                                                              long long ldu, int uw, const gf2k_elim_state *st,
                                                              const u64 *__restrict__ ptab, u64 *__restrict__ tmp,
                                                              long long tld) {
  const int q = blockIdx.x;
  if (q >= st->nmoves) return;
  const long long src = st->mv_src[q];
  const int piv = st->mv_piv[q];
  const long long nA = aw - c0w;
  for (long long w = threadIdx.x; w < nA + uw; w += 256) {
    u64 v;
    if (w < nA)
      v = (piv >= 0 && w < sw) ? ptab[piv * 64 + w] : A[src * lda + c0w + w];
    else
      v = piv >= 0 ? ptab[piv * 64 + sw + (w - nA)] : U[src * ldu + (w - nA)];
    tmp[q * tld + w] = v;
  }
}

__global__ __launch_bounds__(256) void gf2_elim_scatter_kernel(u64 *__restrict__ A, long long lda, long long aw,
                                                               long long c0w, u64 *__restrict__ U, long long ldu, int uw,
                                                               const gf2k_elim_state *st, const u64 *__restrict__ tmp,
                                                               long long tld) {
  const int q = blockIdx.x;
  if (q >= st->nmoves) return;
  const long long dst = st->mv_dst[q];
  const long long nA = aw - c0w;
  for (long long w = threadIdx.x; w < nA + uw; w += 256) {
    const u64 v = tmp[q * tld + w];
    if (w < nA)
      A[dst * lda + c0w + w] = v;
    else
      U[dst * ldu + (w - nA)] = v;
  }
}

__global__ void gf2_elim_begin_block_kernel(gf2k_elim_state *st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) st->r0 = st->r_cur;
}

// pivot row j of the block holds its own original content on the right-hand columns: U'[r0+j][j] ^= 1 turns
// "A ^= U' * P" into "A = U * P" for those rows
__global__ __launch_bounds__(256) void gf2_elim_toggle_kernel(u64 *__restrict__ U, long long ldu, const gf2k_elim_state *st) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int r0 = st->r0;
  if (j < st->r_cur - r0) U[(long long)(r0 + j) * ldu + (j >> 6)] ^= 1ull << (j & 63);
}

// M[i][col0 + i] = 1 for i < n (identity block of an augmented matrix); col0 + i addressed in bits
__global__ __launch_bounds__(256) void gf2_set_diag_kernel(u64 *__restrict__ M, long long ld, int n, long long col0) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const long long c = col0 + i;
    M[(long long)i * ld + (c >> 6)] |= 1ull << (c & 63);
  }
}

// X[pivcols[k]] = R[k] for k < rank (rows of the solution addressed by pivot column); X is zeroed by the caller
__global__ __launch_bounds__(256) void gf2_scatter_rows_kernel(u64 *__restrict__ X, long long ldx, const u64 *__restrict__ R,
                                                               long long ldr, int words, const int *__restrict__ pivcols,
                                                               int rank) {
  const int k = blockIdx.x;
  if (k >= rank) return;
  const long long d = pivcols[k];
  for (int w = threadIdx.x; w < words; w += 256) X[d * ldx + w] = R[(long long)k * ldr + w];
}

// flag = 1 if any word of rows [row_lo, rows) x words [0, words) is non-zero
__global__ __launch_bounds__(256) void gf2_any_nonzero_kernel(const u64 *__restrict__ M, long long ld, int row_lo, int rows,
                                                              int words, int *flag) {
  const long long total = (long long)(rows - row_lo) * words;
  bool nz = false;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
    nz |= M[(row_lo + i / words) * ld + i % words] != 0;
  if (__ballot(nz) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
extern "C" hipError_t gf2k_elim_begin_block(gf2k_elim_state *st, hipStream_t s) {
  hipLaunchKernelGGL(gf2_elim_begin_block_kernel, dim3(1), dim3(64), 0, s, st);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_elim_step(u64 *A, long long lda, int m, long long aw, long long c0w, int sw, int j, u64 colmask,
                                     int full, u64 *U, long long ldu, int uw, gf2k_elim_state *st, int *pivcols, u64 *ptab,
                                     u64 *tmp, long long tld, hipStream_t s) {
  if (sw + uw > 64 || j >= sw) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gf2_elim_pivot_kernel, dim3(1), dim3(1024), 0, s, A, lda, m, c0w + j, colmask, st, pivcols);
  hipLaunchKernelGGL(gf2_elim_ptab_kernel, dim3(64), dim3(64), 0, s, A, lda, c0w, sw, U, ldu, uw, st, ptab);
  const int groups = (m + 3) / 4;
  int grid = (groups + 3) / 4;
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(gf2_elim_update_kernel, dim3(grid), dim3(256), 0, s, A, lda, m, full, c0w, sw, j, U, ldu, uw, st, ptab);
  hipLaunchKernelGGL(gf2_elim_gather_kernel, dim3(128), dim3(256), 0, s, A, lda, aw, c0w, sw, U, ldu, uw, st, ptab, tmp, tld);
  hipLaunchKernelGGL(gf2_elim_scatter_kernel, dim3(128), dim3(256), 0, s, A, lda, aw, c0w, U, ldu, uw, st, tmp, tld);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_elim_toggle(u64 *U, long long ldu, int max_rank, gf2k_elim_state *st, hipStream_t s) {
  if (max_rank <= 0) return hipSuccess;
  hipLaunchKernelGGL(gf2_elim_toggle_kernel, dim3((max_rank + 255) / 256), dim3(256), 0, s, U, ldu, st);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_set_diag(u64 *M, long long ld, int n, long long col0, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gf2_set_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, s, M, ld, n, col0);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_scatter_rows(u64 *X, long long ldx, const u64 *R, long long ldr, int words, const int *pivcols,
                                        int rank, hipStream_t s) {
  if (rank <= 0 || words <= 0) return hipSuccess;
  hipLaunchKernelGGL(gf2_scatter_rows_kernel, dim3(rank), dim3(256), 0, s, X, ldx, R, ldr, words, pivcols, rank);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_any_nonzero(const u64 *M, long long ld, int row_lo, int rows, int words, int *flag, hipStream_t s) {
  if (rows <= row_lo || words <= 0) return hipSuccess;
  long long total = (long long)(rows - row_lo) * words;
  long long grid = (total + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(gf2_any_nonzero_kernel, dim3((unsigned)grid), dim3(256), 0, s, M, ld, row_lo, rows, words, flag);
  return hipGetLastError();
}
