/*
 * m4ri_hip.h -- C ABI of libm4ri_hip.so: the MI355X (gfx950) replacement for the part of
 * the M4RI C library that thomwiggers/m4ri-rust reaches on its multiply path.
 *
 * Section 1 is the DROP-IN boundary: the struct layout and the symbols that m4ri-sys declares
 * in `extern "C"` blocks and that the friendly layer (BinMatrix / BinVector) calls.  Every
 * declaration cites the reference interface it replaces (paths relative to the reference
 * repository root).  Host code that links m4ri-sys against this library instead of libm4ri.a
 * needs no source change: see INTEGRATION.md.
 *
 * Section 2 is the device-resident API (raw device pointers, explicit HIP stream) used by
 * bench.py, the multi-GPU host layer and callers that chain products without the PCIe round
 * trip.  No torch types appear in any signature.
 *
 * All products are computed on the GPU by hand-written HIP kernels; there is no CPU fallback.
 * If no HIP device is usable the multiply entry points print a diagnostic on stderr and return
 * NULL (the m4ri-rust wrapper turns that into its "Multiplication failed" panic,
 * m4ri-rust/src/friendly/binary_matrix.rs:467-469).
 */
#ifndef M4RI_HIP_H
#define M4RI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ===================================================================================== */
/* 1. M4RI-compatible boundary                                                            */
/* ===================================================================================== */

/* m4ri-sys/src/misc.rs:5-26 */
typedef int rci_t;     /* Rci  */
typedef int wi_t;      /* Wi   */
typedef int BIT;       /* BIT  */
typedef uint64_t word; /* Word */
#define m4ri_radix 64
#define m4ri_one ((word)1)
#define m4ri_ffff ((word)0xffffffffffffffffULL)

/* m4ri-sys/src/mzd.rs:16-21 (MzdBlock) */
typedef struct {
  size_t size; /* bytes in this block */
  word *begin;
  word *end;
} mzd_block_t;

/* m4ri-sys/src/mzd.rs:24-79 (#[repr(C)] Mzd, 64 bytes, size asserted mzd.rs:385).
 * Rust reads nrows/ncols/rows directly (binary_matrix.rs:138,176,286,294) and the private
 * fields in mzd_row / mzd_first_row (mzd.rs:277-313): offsets must not move. */
typedef struct mzd_t {
  rci_t nrows;           /*  0 */
  rci_t ncols;           /*  4 */
  wi_t width;            /*  8  ceil(ncols / 64) */
  wi_t rowstride;        /* 12  words between rows (even-padded when width >= 3) */
  wi_t offset_vector;    /* 16  words from blocks[0].begin to row 0 */
  wi_t row_offset;       /* 20 */
  uint8_t flags;         /* 24  mzd.rs:82-94 */
  uint8_t blockrows_log; /* 25 */
  uint8_t padding[14];   /* 26  (byte 26 holds this library's allocation kind) */
  word high_bitmask;     /* 40  valid bits of the last word of a row */
  mzd_block_t *blocks;   /* 48 */
  word **rows;           /* 56  rows[i] = first word of row i, always a HOST pointer */
} mzd_t;

/* m4ri-sys/src/mzd.rs:82-94 */
#define mzd_flag_nonzero_excess 0x2
#define mzd_flag_windowed_zerooffset 0x4
#define mzd_flag_windowed_zeroexcess 0x8
#define mzd_flag_windowed_ownsblocks 0x10
#define mzd_flag_multiple_blocks 0x20

/* --- container support (host side) --- */
mzd_t *mzd_init(rci_t rows, rci_t cols);                           /* mzd.rs:98  */
void mzd_free(mzd_t *A);                                           /* mzd.rs:102 */
mzd_t *mzd_init_window(mzd_t *M, rci_t lowr, rci_t lowc, rci_t highr, rci_t highc); /* mzd.rs:121-127 */
mzd_t *mzd_copy(mzd_t *dst, mzd_t const *src);                     /* mzd.rs:192 */
int mzd_equal(mzd_t const *A, mzd_t const *B);                     /* mzd.rs:187 */
void mzd_randomize(mzd_t *A);                                      /* mzd.rs:184 */
void mzd_set_ui(mzd_t *A, unsigned int value);                     /* mzd.rs:198 */
mzd_t *mzd_transpose(mzd_t *dst, mzd_t const *A);                  /* mzd.rs:148 */
mzd_t *mzd_add(mzd_t *C, mzd_t const *A, mzd_t const *B);          /* mzd.rs:223 */
mzd_t *mzd_sub(mzd_t *C, mzd_t const *A, mzd_t const *B);          /* mzd.rs:230 */
mzd_t *mzd_concat(mzd_t *C, mzd_t const *A, mzd_t const *B);       /* mzd.rs:195 */
mzd_t *mzd_stack(mzd_t *C, mzd_t const *A, mzd_t const *B);        /* mzd.rs:201 */
mzd_t *mzd_submatrix(mzd_t *S, mzd_t const *M, rci_t lowr, rci_t lowc, rci_t highr, rci_t highc); /* mzd.rs:205-212 */
int mzd_is_zero(mzd_t const *A);                                   /* mzd.rs:233 */
void mzd_row_swap(mzd_t *M, rci_t rowa, rci_t rowb);               /* mzd.rs:130 */
void mzd_copy_row(mzd_t *B, rci_t i, mzd_t const *A, rci_t j);     /* mzd.rs:141 */
void mzd_col_swap(mzd_t *M, rci_t cola, rci_t colb);               /* mzd.rs:144 */
void mzd_row_clear_offset(mzd_t *M, rci_t row, rci_t coloffset);   /* mzd.rs:235-240 */
/* inverse of A by elimination of [A | identity] (identity NULL: built here; inv NULL: allocated); NULL if A is singular. mzd.rs:214-218 */
mzd_t *mzd_invert_naive(mzd_t *inv, mzd_t const *A, mzd_t const *identity);
int m4ri_opt_k(int a, int b, int c);                               /* graycode.rs:56 */
/* T[L[v]] = XOR of the rows r + j of M with bit j of v set, for every k-bit v; Gray-code order; host. brilliantrussian.rs:8-17 */
void mzd_make_table(mzd_t const *M, rci_t r, rci_t c, int k, mzd_t *T, rci_t *L);

/* --- the hot path: every product below runs on the GPU --- */

/* C = A*B, Method of the Four Russians. C may be NULL (allocated). k is M4RI's table-size
 * hint (0 = automatic); the device kernel uses 8-bit tables in LDS whatever k says.
 * brilliantrussian.rs:210-216; caller binary_matrix.rs:59 */
mzd_t *mzd_mul_m4rm(mzd_t *C, mzd_t const *A, mzd_t const *B, int k);
/* C ^= A*B. brilliantrussian.rs:218-224 */
mzd_t *mzd_addmul_m4rm(mzd_t *C, mzd_t const *A, mzd_t const *B, int k);
/* C = A*B, Strassen recursion over M4RM leaves; cutoff = minimal dimension for recursion,
 * 0 = library default. strassen.rs:8-18; caller binary_matrix.rs:72 (the default `*`) */
mzd_t *mzd_mul(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff);
/* C ^= A*B. strassen.rs:20-31 */
mzd_t *mzd_addmul(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff);
/* C = A*B, "naive" entry. mzd.rs:150-152; callers binary_matrix.rs:82 and :427 (mul_slice) */
mzd_t *mzd_mul_naive(mzd_t *C, mzd_t const *A, mzd_t const *B);
/* C ^= A*B. mzd.rs:170-173 */
mzd_t *mzd_addmul_naive(mzd_t *C, mzd_t const *A, mzd_t const *B);
/* C (+)= A * Bt^T with Bt pre-transposed (n x l); clear != 0 overwrites. mzd.rs:154-168 */
mzd_t *_mzd_mul_naive(mzd_t *C, mzd_t const *A, mzd_t const *Bt, int clear);
/* C (+)= v*A, v a 1 x l row. mzd.rs:175-181 */
mzd_t *_mzd_mul_va(mzd_t *C, mzd_t const *v, mzd_t const *A, int clear);

/* -- elimination: blocked Gauss-Jordan on the device, trailing updates through the
 *    multiply kernel.  Each call uploads the matrix, works device-resident, and downloads the result. -- */
/* (Reduced) row echelon form in place; full != 0 -> reduced form (unique), full == 0 -> an upper echelon form with
 * the same pivot columns and row space (row contents then depend on the algorithm, as between M4RI's own variants).
 * Returns the rank.  echelonform.rs:9-16; caller binary_matrix.rs:258-261 (BinMatrix::echelonize, rank :250-252) */
rci_t mzd_echelonize(mzd_t *A, int full);
/* same contract; k (table size of the CPU algorithm) is accepted and ignored. echelonform.rs:29-37 */
rci_t mzd_echelonize_m4ri(mzd_t *A, int full, int k);
/* same contract. echelonform.rs:18-27 */
rci_t mzd_echelonize_pluq(mzd_t *A, int full);
/* dst = src^-1 (dst NULL -> allocated); NULL if src is singular. brilliantrussian.rs:201-208; caller binary_matrix.rs:265-268 */
mzd_t *mzd_inv_m4ri(mzd_t *dst, mzd_t const *src, int k);
/* Solve A X = B.  A is m x n, B has >= max(m, n) rows and holds the right-hand side in its first m rows; on return its
 * first n rows hold X (free variables 0) and the remaining rows are 0; A is overwritten (by its reduced echelon form).
 * Returns 0, or -1 if inconsistency_check != 0 and the system has no solution. solve.rs:12-29; caller binary_matrix.rs:582-586 */
int mzd_solve_left(mzd_t *A, mzd_t *B, int cutoff, int inconsistency_check);

/* ===================================================================================== */
/* 2. Device-resident API                                                                 */
/* ===================================================================================== */

/* A dense bit matrix in device memory: row-major 64-bit words, LSB-first, `ld` words between
 * rows (ld even, base 16-byte aligned), excess bits of the last word zero. */
typedef struct {
  uint64_t *data; /* device pointer */
  int64_t ld;     /* words per row stride */
  int nrows;
  int ncols;
} gf2_dmat;

enum { GF2_ALGO_AUTO = 0, GF2_ALGO_M4RM = 1, GF2_ALGO_STRASSEN = 2, GF2_ALGO_NAIVE = 3 };

/* 0 on success, HIP error code (>0) or -1 (bad arguments) otherwise. */
int gf2_device_count(void);                 /* usable HIP devices (0 if none) */
const char *gf2_last_error(void);           /* thread-local text of the last failure */

/* allocation helpers (pooled hipMalloc on the current device).  gf2_dmat_free has hipFree's semantics: it waits for the
 * owning device (the API above is asynchronous; queued work may still use the block) before the block is recycled.
 * gf2_dmat_free_async recycles the block once everything queued on `stream` SO FAR has completed: for matrices that were
 * only used on that one stream (temporaries of a chain of products). */
int gf2_dmat_alloc(gf2_dmat *M, int nrows, int ncols);
void gf2_dmat_free(gf2_dmat *M);
int gf2_dmat_free_async(gf2_dmat *M, void *stream);
int gf2_dmat_upload(gf2_dmat *dst, mzd_t const *src, void *stream);   /* host mzd_t -> device; returns after the copy */
int gf2_dmat_download(mzd_t *dst, gf2_dmat const *src, void *stream); /* device -> host mzd_t */
int gf2_dmat_fill_random(gf2_dmat *M, uint64_t seed, void *stream);   /* splitmix64 stream, same as the oracle */
/* same stream, but M holds rows [row0, row0 + M->nrows) of the full matrix (row-block shards) */
int gf2_dmat_fill_random_rows(gf2_dmat *M, uint64_t seed, int64_t row0, void *stream);
/* general block: rows [row0, ..) x words [col_word0, ..) of a seeded matrix with full_ncols columns (column-panel shards) */
int gf2_dmat_fill_random_block(gf2_dmat *M, uint64_t seed, int64_t row0, int64_t col_word0, int full_ncols, void *stream);

/* C (+)= A*B on `stream` (hipStream_t, NULL = default stream); asynchronous.
 * algo: GF2_ALGO_*; param: Strassen levels when algo==STRASSEN (0 = automatic), else ignored. */
int gf2_mul_dev(gf2_dmat *C, gf2_dmat const *A, gf2_dmat const *B, int accumulate, int algo, int param,
                void *stream);
/* C (+)= A * Bt^T (row-parity form, mzd.rs:154-168) */
int gf2_mul_nt_dev(gf2_dmat *C, gf2_dmat const *A, gf2_dmat const *Bt, int accumulate, void *stream);
/* C = A xor B (mzd.rs:223), D = S^T (mzd.rs:148), equality (mzd.rs:187; *equal = 1/0) */
int gf2_add_dev(gf2_dmat *C, gf2_dmat const *A, gf2_dmat const *B, void *stream);
int gf2_transpose_dev(gf2_dmat *D, gf2_dmat const *S, void *stream);
int gf2_equal_dev(gf2_dmat const *A, gf2_dmat const *B, int *equal, void *stream);

/* In-place echelon form of the first ncols_limit columns (0 = all) of a device matrix; the remaining columns follow
 * the row operations (augmented systems).  *rank receives the rank, pivot_cols (host, may be NULL, >= min(rows, limit)
 * ints) the pivot columns.  Synchronous. */
int gf2_echelonize_dev(gf2_dmat *A, int full, int ncols_limit, int *rank, int *pivot_cols, void *stream);
/* Ainv = A^-1 for square A; *singular = 1 (Ainv untouched) if A has no inverse.  Synchronous. */
int gf2_inverse_dev(gf2_dmat *Ainv, gf2_dmat const *A, int *singular, void *stream);

/* Strassen levels a product of this shape would use right now (0 = plain M4RM): the cost model's choice, lowered until the
 * operand arena of that many levels fits into free device memory (without a device: the cost model's choice) */
int gf2_strassen_levels(int m, int l, int n, int algo, int param);
/* How a device product of this shape would run, by the cost model (no device needed): returns the Strassen level count;
 * *kind = 0 the shape as given, 1 zero-padded up to dims[0..2], 2 peeled down to the core dims[0..2] (border strips plain) */
int gf2_mul_plan(int m, int l, int n, int algo, int param, int *kind, int dims[3]);
/* How ONE (batched) tile-kernel launch of `batch` products m x l x n would run, by the cost model (no device needed); `packed`:
 * A is handed over row-group packed.  out[0] = kernel variant (7 / 20: 1024- / 256-row tiles x 2048 columns; 8: 2048 x 1024;
 * 9 / 10 / 11 / 12: 4096 / 2048 / 1024 / 512 rows x 512 columns), out[1] = uniform slices of the inner dimension (variants 7, 8,
 * 20), out[2] = tiles cut into stream-K segments and out[3] = number of segments (variants 9-12), out[4] = bytes of scratch
 * for partial tiles; a batched launch may be cut in two -- out[5] = products in the second launch (0: one launch), out[6] / out[7] /
 * out[8] = its variant, tiles cut and segments; returns the modelled time in seconds.  Diagnostic: tools and tests read the
 * launcher's choice from here. */
double gf2_tile_plan(int m, int l, int n, int batch, int packed, long long out[9]);
/* ... and the plan's row band, if it has one: the rows below the last whole tile row of the launch(es) above run in a launch of
 * their own with a shorter tile.  out[0] = rows of the band (0: none; gf2_tile_plan then describes all rows), out[1] = its variant,
 * out[2] / out[3] = its tiles cut and segments, out[4] = its scratch bytes (gf2_tile_plan's out[4] already covers them). */
void gf2_tile_plan_band(int m, int l, int n, int batch, int packed, long long out[5]);
/* modelled seconds of a device product of this shape with exactly `levels` Strassen levels (0: plain M4RM); -1 if that many
 * levels do not divide the shape.  Diagnostic (tools/levels_sweep.py prints it beside the measured time). */
double gf2_model_time(int m, int l, int n, int levels);
/* bytes the Strassen split / merge passes of a product with that many levels read and write (0 for levels == 0): every pass
 * kernel reads its sources once and writes its destinations once, so this is exact; bench.py prices the passes with it */
double gf2_strassen_pass_bytes(int m, int l, int n, int levels);
/* bytes of scratch a product of this shape needs (Strassen operands, packed copies);
 * the library keeps one cached arena per device and grows it on demand. */
size_t gf2_mul_workspace_bytes(int m, int l, int n, int algo, int param);

/* One process, several devices (north_star: "shard row-blocks of A across the GPUs of one node"): C = A*B on host
 * matrices with the rows of A and C divided among `devices` (ndev ordinals in hipGetDeviceCount's numbering; an ordinal may
 * repeat = several shares on one device).  Every share uploads its rows of A and its own copy of B over its own PCIe
 * link, multiplies and downloads its rows of C; the inner dimension is never split, so there is no reduction.  C NULL:
 * allocated.  Returns C, or NULL on failure.  algo / param as gf2_mul_dev.
 * The drop-in entry points (mzd_mul, mzd_mul_m4rm, mzd_mul_naive, strassen.rs:18 / brilliantrussian.rs:216 / mzd.rs:152) do
 * the same by themselves when asked to: M4RI_HIP_DEVICES = "auto" (more than one device visible and the product large: >= 7e13
 * bit operations, >= 4096 rows per share; ignored when WORLD_SIZE > 1, i.e. inside a torch.distributed job) | "all" | "0,1,...".
 * Unset: the current device only.  A single ordinal pins every host entry point (products, elimination, transpose, operand
 * cache) to that device. */
mzd_t *gf2_mul_multi(mzd_t *C, mzd_t const *A, mzd_t const *B, int algo, int param, const int *devices, int ndev);

/* Operand cache for the drop-in entry points: keep a device copy of the host matrix M until gf2_mzd_uncache(M) or
 * mzd_free(M); mzd_mul* calls whose A or B is M then skip its upload (repeated A*v with a fixed A -- mul_slice,
 * binary_matrix.rs:416-431 -- is otherwise bound by moving A over PCIe on every call).  The caller promises not to
 * change M's bits through the host pointers meanwhile (mzd_write_bit and friends on the Rust side are plain stores the
 * library cannot see); library calls that write M or a window of M (as destination, mzd_echelonize, mzd_solve_left, ...)
 * drop the copy themselves: the cache is keyed by the block that M and its windows share. */
int gf2_mzd_cache_on_device(mzd_t const *M);
void gf2_mzd_uncache(mzd_t const *M);

/* Host blocks of at least M4RI_HIP_PIN_MIN_BYTES (1 MiB) are pinned so that uploads and downloads run at the PCIe rate; pinning
 * fresh pages costs about 100 ms per 512 MiB, so freed blocks are pooled (up to M4RI_HIP_PIN_CACHE_BYTES, 8 GiB).  A caller that
 * knows its sizes can fill the pool ahead of time: gf2_mzd_prewarm(r, c, count) creates `count` pinned blocks of the size of an
 * r x c matrix (returns how many it added).  The first mzd_init / mzd_mul(NULL, ...) / mzd_transpose(NULL, ...) of that size is
 * then as fast as the later ones. */
int gf2_mzd_prewarm(rci_t r, rci_t c, int count);

/* give cached device memory (per-stream scratch arenas, block cache) of the current device back to the driver;
 * waits for the device first.  The library otherwise keeps what it allocated: the arena of a 131072^3 product is 141 GiB. */
int gf2_trim(void);

/* Size dispatch of the drop-in entry points.  Products of at most M4RI_HIP_HOST_SMALL_WORK
 * word operations (m * l * ceil(n / 64); default 2^20, 0 = every product goes to the device) and echelon forms of that size
 * are computed on the host by the library's own word-parallel routines: the reference's bench shapes (10 x 10 ... 1000 x 64 x
 * 1000, benches/binary_matrix.rs:30-76) take less time there than one PCIe round trip.  The entry points still require a
 * usable device (no fallback: without one they fail as before).  The routines are exported so that they can be checked
 * against the oracle without a device; gf2_host_small_calls() counts how often they ran. */
int gf2_mul_host_small(mzd_t *C, mzd_t const *A, mzd_t const *B, int accumulate);      /* C (+)= A*B; 0 on success */
int gf2_mul_nt_host_small(mzd_t *C, mzd_t const *A, mzd_t const *Bt, int accumulate);  /* C (+)= A*Bt^T */
int gf2_echelonize_host_small(mzd_t *A, int full);                                      /* in place; returns the rank */
long long gf2_host_small_calls(void);

/* compact binary file format for host matrices ("GF2M", version, nrows, ncols, dense little-endian rows);
 * 0 / non-NULL on success.  (the reference only serialises to JSON, binary_matrix.rs:10-35) */
int gf2_mzd_save(const char *path, mzd_t const *M);
mzd_t *gf2_mzd_load(const char *path);

/* kernel timing for bench.py's roofline object: HIP events recorded on the launch stream
 * around every launch of the dominant multiply kernel while enabled. */
void gf2_prof_enable(int on);
/* sums since the last reset: *launches, *ms = total event-measured duration of those launches */
int gf2_prof_read(int *launches, double *ms, int reset);

/* Schedules of a large product on HOST matrices (upload / multiply / download pipelines), as the library's time model plays them
 * through: t_end[i] = modelled seconds of schedule i + 1 in M4RI_HIP_HOST_PLAN's numbering, -1 where a schedule does not apply:
 *   1 row blocks of A and C;  2 / 3 / 4 slabs of the inner dimension (four equal; 1/8 1/8 1/4 1/2; two halves), the last slab in row
 *   blocks;  5 / 6 two row groups, each through four equal slabs / two halves;  7 / 8 / 12 the first row group through slabs that grow
 *   from 1/16 (a short lead-in while B arrives), the second through two halves / four quarters / the whole inner dimension;  9 / 10 the
 *   first group through 1/8 1/8 1/4 1/2 / four equal slabs, the second through two halves;  11 one group through the growing slabs.
 * Returns the number of the schedule mzd_mul takes for this shape (0: not pipelined). */
int gf2_host_plan_model(int m, int l, int n, int algo, int param, double t_end[12]);

/* Launch census: "<count> <mangled kernel name>\n" for every kernel of the library this process has launched so far, written to buf
 * (at most cap - 1 bytes and a terminator); returns the length of the whole text.  With M4RI_HIP_KERNEL_CENSUS_FILE=<path> in the
 * environment the counts are also appended to that file when the library is unloaded (child processes of a test suite).  The GPU
 * suite ends with a test that every kernel the shared object contains has been launched (tests/test_zz_kernel_census.py). */
size_t gf2_kernel_census(char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* M4RI_HIP_H */
