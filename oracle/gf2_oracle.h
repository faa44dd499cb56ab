/*
 * gf2_oracle.h -- CPU restatement of the M4RI multiply path reached by m4ri-rust.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS: the arithmetic of the reference lives in the third-party M4RI C
 * library (git submodule m4ri-sys/vendor/m4ri, https://bitbucket.org/thomwiggers/m4ri.git,
 * NO pinned revision: the submodule directory is empty and Cargo.lock is git-ignored),
 * so it can be neither compiled nor run here.  This file restates M4RI's *published*
 * algorithms (naive AND/parity product, Method of the Four Russians with Gray-code
 * tables, Strassen-Winograd) under the memory convention the reference pins:
 *   bit j of row i = bit (j % 64), LSB first, of word (j / 64) of the row
 *   (m4ri-sys/src/mzd.rs:246-269), excess bits of the last word zero
 *   (m4ri-rust/src/friendly/binary_matrix.rs:151-155).
 * It is pinned against the only known-answer tests the reference holds for the path
 * (identity products and vector products, binary_matrix.rs:662-686; bit order,
 * mzd.rs:425-460; serde word order, binary_matrix.rs:695-699) and against
 * tests/golden/ vectors produced by an independent numpy bit-level product.
 * For any non-identity product: PARITY UNPINNED by the reference's own fixtures
 * (none exist); it rests on the exactness/uniqueness of the GF(2) product.
 *
 * All matrices here are dense row-major arrays of 64-bit words with a row stride
 * given in words ("ld").
 */
#ifndef GF2_ORACLE_H
#define GF2_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* splitmix64 fill: word t of the stream with the given seed (input generator shared by
 * tests, golden fixtures and bench; stands in for mzd_randomize, mzd.rs:183-184). */
uint64_t oracle_splitmix64(uint64_t seed, uint64_t t);
void oracle_fill_random(uint64_t *M, int nrows, int ncols, int ld, uint64_t seed);

/* bit-level triple loop: the obviously-correct definition of C = A*B over GF(2). */
void oracle_mul_bits(uint64_t *C, int ldc, const uint64_t *A, int lda, const uint64_t *B, int ldb,
                     int m, int l, int n);

/* _mzd_mul_naive (mzd.rs:154-168): C (+)= A * Bt^T, Bt pre-transposed (n x l):
 * C[i][j] = parity(popcount(A.row(i) AND Bt.row(j))). clear!=0 overwrites C. */
void oracle_mul_naive_t(uint64_t *C, int ldc, const uint64_t *A, int lda, const uint64_t *Bt, int ldbt,
                        int m, int l, int n, int clear);

/* mzd_mul_naive (mzd.rs:150-152): transposes B, then the routine above. */
void oracle_mul_naive(uint64_t *C, int ldc, const uint64_t *A, int lda, const uint64_t *B, int ldb,
                      int m, int l, int n);

/* m4ri_opt_k (graycode.rs:44-56): k ~ 0.75*log2(n). */
int oracle_opt_k(int a, int b, int c);

/* mzd_make_table (brilliantrussian.rs:8-17): all 2^k XOR combinations of rows r..r+k-1 of M,
 * generated in Gray-code order; T is 2^k x ncols, L[v] = row of T holding combination v. */
void oracle_make_table(const uint64_t *M, int ldm, int ncols, int r, int k, uint64_t *T, int ldt, int *L);

/* mzd_mul_m4rm / mzd_addmul_m4rm (brilliantrussian.rs:210-224): k==0 -> oracle_opt_k.
 * clear!=0: C = A*B, else C ^= A*B. */
void oracle_mul_m4rm(uint64_t *C, int ldc, const uint64_t *A, int lda, const uint64_t *B, int ldb,
                     int m, int l, int n, int k, int clear);

/* mzd_mul / mzd_addmul (strassen.rs:8-31): Strassen-Winograd, recursion while every
 * dimension >= 2*cutoff, M4RM leaves. cutoff==0 -> default. */
void oracle_mul_strassen(uint64_t *C, int ldc, const uint64_t *A, int lda, const uint64_t *B, int ldb,
                         int m, int l, int n, int cutoff, int clear);

/* _mzd_mul_va (mzd.rs:175-181): C(1 x n) (+)= v(1 x l) * A(l x n). */
void oracle_mul_va(uint64_t *C, const uint64_t *v, const uint64_t *A, int lda, int l, int n, int clear);

/* mzd_transpose (mzd.rs:146-148) and mzd_add (mzd.rs:220-223). */
void oracle_transpose(uint64_t *D, int ldd, const uint64_t *S, int lds, int nrows, int ncols);
void oracle_add(uint64_t *C, int ldc, const uint64_t *A, int lda, const uint64_t *B, int ldb, int nrows, int ncols);

/* tuned single-thread CPU baseline ("port"): 8-bit tables, 4 tables per pass, Strassen on top.
 * Same result as oracle_mul_strassen; timed by bench.py's cpu_baseline leg. */
void oracle_mul_fast(uint64_t *C, int ldc, const uint64_t *A, int lda, const uint64_t *B, int ldb,
                     int m, int l, int n);

/* ---- elimination ------------------------------------------------
 * Textbook Gauss(-Jordan) over GF(2), the semantics documented at echelonform.rs:9-16: pivots are searched
 * column by column, the first row at or below the current rank with a 1 is swapped up.  full != 0 clears the
 * pivot column in every other row (reduced row echelon form: unique, so any correct implementation agrees bit for
 * bit); full == 0 clears it below only (the row contents then depend on the algorithm, M4RI's variants differ
 * among themselves; only rank, pivot columns and row space are comparable).
 * Only the first `limit` columns (0 = all) are searched for pivots; the others follow the row operations.
 * pivcols (may be NULL) receives the pivot columns.  Returns the rank.
 * PARITY UNPINNED by the reference: it holds no test for rank/echelonize/inverted/solve_left. */
int oracle_echelonize(uint64_t *M, int ld, int nrows, int ncols, int limit, int full, int *pivcols);
/* mzd_inv_m4ri (brilliantrussian.rs:201-208): Ainv = A^-1, returns 0; returns -1 if A is singular. */
int oracle_inverse(uint64_t *Ainv, int ldi, const uint64_t *A, int lda, int n);
/* mzd_solve_left (solve.rs:12-29) for A (m x n), B (brows >= max(m,n) rows, k columns, right-hand side in the
 * first m rows): B's first n rows become X with free variables 0, the rest 0.  Returns 0, or -1 if inconsistent. */
int oracle_solve_left(const uint64_t *A, int lda, int m, int n, uint64_t *B, int ldb, int brows, int k);

#ifdef __cplusplus
}
#endif
#endif
