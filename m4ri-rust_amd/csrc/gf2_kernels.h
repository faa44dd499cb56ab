// gf2_kernels.h -- internal interface between the kernels (gf2_kernels.hip) and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// One (batched) product C (+)= A*B on dense device buffers; strides in 64-bit words.
// (tests/test_gpu_kernel_variants.py mirrors this struct with ctypes: keep the two in step.)
struct gf2k_mul_args {
  const uint64_t *A;
  const uint64_t *B;
  uint64_t *C;
  long long lda, ldb, ldc;  // row strides
  long long sA, sB, sC;     // batch strides
  int m, l, n;              // bits
  int tiles_m, tiles_n;     // filled in by the launcher
  int ksplit;               // slices of the inner dimension (<= 1: none); combined with atomic XOR
  int kwords;               // 32-bit words of the inner dimension per slice (filled in by the launcher)
  int batch;
  int accumulate;  // 0: C = A*B, 1: C ^= A*B
  const uint32_t *Bp;  // chunk-packed copy of B (kernel variants with BPACK), else unused
  long long sBp;       // batch stride of Bp in 2 KiB blocks
  int bp_nc;           // chunk blocks per tile column in Bp
  // split-K without atomics: slice ks of batch item bt stores its partial product at P + (bt*ksplit + ks)*sP (dense,
  // row stride ldp) and gf2k_m4rm combines the slices into C with a second kernel; nullptr: atomic XOR into C
  uint64_t *P;
  long long ldp, sP;
  // A is stored row-group packed (gf2k_strassen_split2 with side 2): u64 index ((r / 64) * lda + c) * 64 + r % 64, ceil(m / 64) * 64 rows
  int a_packed;
  // stream-K split of a v8 launch (cfg 9-12; gf2_kernels.hip, gf2_m4rm_kernel_v8).  Caller: n_rem = tiles (counted from the END
  // of the tile order) to be cut into segments, nseg = how many segments it would like (0: 256), P / p_words = scratch for
  // the partial tiles (gf2k_m4rm_streamk_words).  The launcher fills in the rest; n_rem = 0 or P = nullptr: whole tiles only.
  int n_rem, nseg;
  long long p_words;  // capacity of P in 64-bit words
  int n_full, seg_slabs, tile_slabs;  // (launcher) whole-tile workgroups; 64-bit slabs per segment / per tile
};

// Device-side record of a blocked elimination (gf2_elim.hip): the kernels of a step read their ranges from it, so a
// whole column block is enqueued without host round trips.
struct gf2k_elim_state {
  int r0;      // rank when the current column block started: rows [0, r0) hold the pivots of the earlier blocks in order
  int r_cur;   // rank so far
  int np;      // pivots found by the current step (one 64-bit word of columns)
  int nmoves;  // row moves of the block-end permutation
  int jbase;   // r_cur - r0 before the current step: index of its first pivot inside the block
  int scan;    // every row in [r0, scan) is a pivot of the current block: the search for candidates starts here
  int lastword;  // after gf2k_elim_end_block: last word index (absolute) in which a pivot row of the block is non-zero
  unsigned long long pcmask;  // pivot columns of the current word
  int cur_row[64];            // rows chosen by the current step (their flag says "pivot of this step" until the next one)
  int cntP, cnt2;             // look-ahead launches, counted up through a block: update workgroups whose PRIORITY rows have landed / that have finished
  int err;                    // set when the look-ahead workgroup's bounded wait ran out (never expected)
};
// Row flags (one byte per row) during a block: 0 = ordinary row, 1 + c = pivot of the CURRENT step with pivot column c of
// the word, 255 = pivot of an earlier step of this block.  Rows are not moved inside a block; gf2k_elim_end_block brings
// block pivot j to row r0 + j.
enum { GF2K_ELIM_BLOCK_PIVOTS = 2048 };

extern "C" {
hipError_t gf2k_elim_begin_block(gf2k_elim_state *st, hipStream_t s);
// one 64-column step.  ptab: 64 x 64 + 256 + 512 words (the step's raw pivot rows over the block, the selector map, the stash of the
// next search: 256 words and 256 flags); the steps of a block must be enqueued in order j = 0 .. sw - 1 on one stream with the same
// `lookahead` (the stash a step leaves behind is read by the next step's launch)
hipError_t gf2k_elim_step(uint64_t *A, long long lda, int m, long long c0w, int sw, int j, uint64_t colmask, int full,
                          uint64_t *U, long long ldu, int uw, gf2k_elim_state *st, int *pivcols, uint64_t *ptab,
                          unsigned char *rowflag, int *blkpiv, uint64_t colmask_next, int lookahead, hipStream_t s);
// end of a block: permutation of whole rows (columns [c0w, aw) and the tracking words) through `tmp` (>= 2 *
// GF2K_ELIM_BLOCK_PIVOTS rows of tld words), flags cleared, U' toggled, st->lastword set over words [w_right, aw);
// `moves` holds 4 * GF2K_ELIM_BLOCK_PIVOTS ints
hipError_t gf2k_elim_end_block(uint64_t *A, long long lda, long long aw, long long c0w, uint64_t *U, long long ldu, int uw,
                               gf2k_elim_state *st, unsigned char *rowflag, const int *blkpiv, int *moves, uint64_t *tmp,
                               long long tld, long long w_right, hipStream_t s);
// whole-matrix-in-LDS elimination for small matrices; hipErrorInvalidValue if the matrix does not qualify
hipError_t gf2k_elim_small(uint64_t *A, long long lda, int m, int ncols, int limit, int full, int *rank_out, int *pivcols,
                           hipStream_t s);
hipError_t gf2k_set_diag(uint64_t *M, long long ld, int n, long long col0, hipStream_t s);
hipError_t gf2k_scatter_rows(uint64_t *X, long long ldx, const uint64_t *R, long long ldr, int words, const int *pivcols,
                             int rank, hipStream_t s);
hipError_t gf2k_any_nonzero(const uint64_t *M, long long ld, int row_lo, int rows, int words, int *flag, hipStream_t s);
int gf2k_m4rm_rows_per_tile(int cfg);
// 64-bit words of partial-tile scratch a stream-K launch of variant `cfg` with `nseg` segments needs (two slots per segment)
long long gf2k_m4rm_streamk_words(int cfg, int nseg);
int gf2k_m4rm_cols_per_tile(int cfg);
// row-group-packed copy of A for gf2k_mul_args::a_packed: dst holds ceil(m/64)*64 rows of wp (even, >= w) words
hipError_t gf2k_packA(uint64_t *dst, long long wp, const uint64_t *src, long long lds_, int m, int w, hipStream_t stream);
hipError_t gf2k_m4rm(gf2k_mul_args a, int cfg, hipStream_t stream);
#ifdef GF2K_DEV_VARIANTS  // development builds only (tools/libm4ri_hip_dev.so): diagnostics and the chunk-packed B experiment
hipError_t gf2k_dbg_sec(unsigned long long *out8);
int gf2k_packB_chunks(int l);
hipError_t gf2k_packB(uint32_t *Bp, long long bpStride, const uint64_t *B, long long ldb, long long bStride, int l, int n,
                      int batch, hipStream_t stream);
#endif
hipError_t gf2k_rowparity(const uint64_t *A, long long lda, const uint64_t *Bt, long long ldbt, uint64_t *C, long long ldc,
                          int m, int l, int n, int accumulate, hipStream_t stream);
hipError_t gf2k_narrow(const uint64_t *A, long long lda, const uint64_t *B, long long ldb, uint64_t *C, long long ldc,
                       int m, int l, int n, int accumulate, hipStream_t stream);
// n <= 32 vectors against a long inner dimension: a wave per row (Bt: n rows of l bits)
hipError_t gf2k_widevec(const uint64_t *A, long long lda, const uint64_t *Bt, long long ldbt, uint64_t *C, long long ldc,
                        int m, int l, int n, int accumulate, int jshift, hipStream_t stream);
hipError_t gf2k_tallskinny(const uint64_t *A, long long lda, const uint64_t *B, long long ldb, uint64_t *C, long long ldc,
                           int m, int l, int n, int accumulate, hipStream_t stream);
// the same for one to four vectors against rows of 65..256 bits, with the packed transposed product written beside C (word
// side[j * side_ld + row / 64] = bit j of rows row .. row + 63); A, C and side may be pinned HOST memory (the kernel then streams
// them over PCIe itself: no copies); hipErrorNotSupported for other shapes
hipError_t gf2k_tallskinny_side(const uint64_t *A, long long lda, const uint64_t *B, long long ldb, uint64_t *C, long long ldc, int m,
                                int l, int n, uint64_t *side, long long side_ld, hipStream_t stream);
// n <= 64 vectors, any inner dimension: 4-bit tables per 512-bit slab, inner dimension divided among workgroups (atomic XOR into C)
hipError_t gf2k_tallskinny_long(const uint64_t *A, long long lda, const uint64_t *B, long long ldb, uint64_t *C, long long ldc,
                                int m, int l, int n, int accumulate, hipStream_t stream);
hipError_t gf2k_va(const uint64_t *A, long long lda, const uint64_t *B, long long ldb, uint64_t *C, long long ldc, int m,
                   int l, int n, hipStream_t stream);
hipError_t gf2k_xor2d(uint64_t *C, long long ldc, const uint64_t *A, long long lda, const uint64_t *B, long long ldb,
                      int rows, int words, hipStream_t stream);
hipError_t gf2k_padcopy(uint64_t *dst, long long ldd, int drows, int dwords, const uint64_t *src, long long lds_, int srows,
                        int swords, hipStream_t stream);
hipError_t gf2k_fill_random(uint64_t *M, long long ld, int rows, int cols, uint64_t seed, long long row0, long long fullw,
                            long long colw0, hipStream_t stream);
hipError_t gf2k_diff(const uint64_t *A, long long lda, const uint64_t *B, long long ldb, int rows, int cols, int *diff,
                     hipStream_t stream);
hipError_t gf2k_transpose(uint64_t *D, long long ldd, const uint64_t *S, long long lds_, int rows, int cols,
                          hipStream_t stream);
hipError_t gf2k_strassen_split(uint64_t *dst, long long ldd, long long dstStride, const uint64_t *src, long long lds_,
                               long long srcStride, int h, int w, int side, int batch, hipStream_t stream);
hipError_t gf2k_strassen_split2(uint64_t *dst, long long ldd, long long dstStride, const uint64_t *src, long long lds_,
                                long long srcStride, int h, int w, int side, int batch, hipStream_t stream);
hipError_t gf2k_strassen_split3(uint64_t *dst, long long ldd, long long dstStride, const uint64_t *const *src0,
                                const uint64_t *const *src1, int groups, long long lds_, long long srcStride, int h, int w, int side,
                                int batch, hipStream_t stream);
hipError_t gf2k_strassen_merge3(uint64_t *dst, long long ldd, long long dstStride, const uint64_t *src, long long lds_,
                                long long srcStride, int h, int w, int accumulate, int groups, int batch, hipStream_t stream);
hipError_t gf2k_strassen_merge2(uint64_t *dst, long long ldd, long long dstStride, const uint64_t *src, long long lds_,
                                long long srcStride, int h, int w, int accumulate, int batch, hipStream_t stream);
hipError_t gf2k_strassen_merge(uint64_t *dst, long long ldd, long long dstStride, const uint64_t *src, long long lds_,
                               long long srcStride, int h, int w, int accumulate, int batch, hipStream_t stream);
}

// Launch census (include/m4ri_hip.h, gf2_kernel_census): every kernel launch of the library goes through this macro, which counts
// it by the kernel's host-side handle before handing it to HIP's own form.  tests/test_zz_kernel_census.py compares the kernels that
// were LAUNCHED by the GPU suite with the kernels the shared object CONTAINS (round 4: a kernel nobody launched ran wrong for half a round).
void gf2k_note_launch(const void *kernel);
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...)                        \
  do {                                                             \
    gf2k_note_launch(reinterpret_cast<const void *>(kernelName));  \
    hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__);         \
  } while (0)
