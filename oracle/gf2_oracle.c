/*
 * gf2_oracle.c -- CPU restatement of the M4RI multiply path (see gf2_oracle.h header:
 * TEST INFRASTRUCTURE ONLY; parity unpinned by reference fixtures for non-identity products).
 *
 * Each function cites the reference declaration (relative to /root/reference) whose
 * documented semantics it follows.  The M4RI C sources themselves are absent from the
 * reference tree (empty submodule m4ri-sys/vendor/m4ri), so what is restated is the
 * published algorithm, not a line-by-line port.
 */
#include "gf2_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;

static inline int width_of(int ncols) { return (ncols + 63) / 64; }
static inline u64 tail_mask(int ncols) { return (ncols % 64) ? ((1ULL << (ncols % 64)) - 1) : ~0ULL; }

/* ---- input generator ---------------------------------------------------------------- */

u64 oracle_splitmix64(u64 seed, u64 t) {
  u64 z = seed + (t + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* stands in for mzd_randomize (m4ri-sys/src/mzd.rs:183-184): uniform bits, excess bits zero */
void oracle_fill_random(u64 *M, int nrows, int ncols, int ld, u64 seed) {
  int w = width_of(ncols);
  u64 tm = tail_mask(ncols);
  for (int i = 0; i < nrows; ++i)
    for (int j = 0; j < w; ++j) {
      u64 v = oracle_splitmix64(seed, (u64)i * (u64)w + (u64)j);
      if (j == w - 1) v &= tm;
      M[(size_t)i * ld + j] = v;
    }
}

static inline int get_bit(const u64 *row, int j) { return (int)((row[j >> 6] >> (j & 63)) & 1); }

/* read kk (<=32) bits of a row starting at bit r; bits beyond ncols must already be zero */
static inline unsigned read_bits(const u64 *row, int r, int kk) {
  int w = r >> 6, s = r & 63;
  u64 v = row[w] >> s;
  if (s + kk > 64) v |= row[w + 1] << (64 - s);
  return (unsigned)(v & ((kk >= 32) ? 0xFFFFFFFFULL : ((1ULL << kk) - 1)));
}

static void zero_rows(u64 *C, int ldc, int m, int n) {
  int w = width_of(n);
  for (int i = 0; i < m; ++i) memset(C + (size_t)i * ldc, 0, (size_t)w * sizeof(u64));
}

/* ---- definition --------------------------------------------------------------------- */

void oracle_mul_bits(u64 *C, int ldc, const u64 *A, int lda, const u64 *B, int ldb, int m, int l, int n) {
  zero_rows(C, ldc, m, n);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) {
      int acc = 0;
      for (int t = 0; t < l; ++t) acc ^= get_bit(A + (size_t)i * lda, t) & get_bit(B + (size_t)t * ldb, j);
      if (acc) C[(size_t)i * ldc + (j >> 6)] |= 1ULL << (j & 63);
    }
}

/* ---- mzd_transpose (m4ri-sys/src/mzd.rs:146-148) ------------------------------------- */

void oracle_transpose(u64 *D, int ldd, const u64 *S, int lds, int nrows, int ncols) {
  zero_rows(D, ldd, ncols, nrows);
  for (int i = 0; i < nrows; ++i)
    for (int j = 0; j < ncols; ++j)
      if (get_bit(S + (size_t)i * lds, j)) D[(size_t)j * ldd + (i >> 6)] |= 1ULL << (i & 63);
}

/* ---- mzd_add (m4ri-sys/src/mzd.rs:220-223): C = A + B = A xor B ----------------------- */

void oracle_add(u64 *C, int ldc, const u64 *A, int lda, const u64 *B, int ldb, int nrows, int ncols) {
  int w = width_of(ncols);
  for (int i = 0; i < nrows; ++i)
    for (int j = 0; j < w; ++j) C[(size_t)i * ldc + j] = A[(size_t)i * lda + j] ^ B[(size_t)i * ldb + j];
}

/* ---- _mzd_mul_naive (m4ri-sys/src/mzd.rs:154-168) ------------------------------------- */

void oracle_mul_naive_t(u64 *C, int ldc, const u64 *A, int lda, const u64 *Bt, int ldbt, int m, int l, int n,
                        int clear) {
  int wl = width_of(l);
  u64 tm = tail_mask(l);
  if (clear) zero_rows(C, ldc, m, n);
  for (int i = 0; i < m; ++i) {
    const u64 *a = A + (size_t)i * lda;
    for (int j = 0; j < n; ++j) {
      const u64 *b = Bt + (size_t)j * ldbt;
      u64 acc = 0;
      for (int t = 0; t < wl; ++t) {
        u64 v = a[t] & b[t];
        if (t == wl - 1) v &= tm;
        acc ^= v;
      }
      if (__builtin_parityll(acc)) C[(size_t)i * ldc + (j >> 6)] ^= 1ULL << (j & 63);
    }
  }
}

/* ---- mzd_mul_naive (m4ri-sys/src/mzd.rs:150-152) -------------------------------------- */

void oracle_mul_naive(u64 *C, int ldc, const u64 *A, int lda, const u64 *B, int ldb, int m, int l, int n) {
  int ldbt = width_of(l);
  u64 *Bt = (u64 *)calloc((size_t)n * ldbt + 1, sizeof(u64));
  oracle_transpose(Bt, ldbt, B, ldb, l, n);
  oracle_mul_naive_t(C, ldc, A, lda, Bt, ldbt, m, l, n, 1);
  free(Bt);
}

/* ---- _mzd_mul_va (m4ri-sys/src/mzd.rs:175-181) ---------------------------------------- */

void oracle_mul_va(u64 *C, const u64 *v, const u64 *A, int lda, int l, int n, int clear) {
  int w = width_of(n);
  if (clear) memset(C, 0, (size_t)w * sizeof(u64));
  for (int t = 0; t < l; ++t)
    if (get_bit(v, t))
      for (int j = 0; j < w; ++j) C[j] ^= A[(size_t)t * lda + j];
}

/* ---- m4ri_opt_k (m4ri-sys/src/graycode.rs:44-56): 0.75*log2(n), n = b for multiply ----- */

int oracle_opt_k(int a, int b, int c) {
  int n = (c != 0) ? b : (a < b ? a : b);
  int lg = 0;
  while ((1 << (lg + 1)) <= n && lg < 30) ++lg;
  int k = (int)(0.75 * (double)(1 + lg));
  if (k < 1) k = 1;
  if (k > 16) k = 16;
  return k;
}

/* ---- mzd_make_table (m4ri-sys/src/brilliantrussian.rs:8-17) ---------------------------- */

void oracle_make_table(const u64 *M, int ldm, int ncols, int r, int k, u64 *T, int ldt, int *L) {
  int w = width_of(ncols);
  memset(T, 0, (size_t)w * sizeof(u64));
  L[0] = 0;
  unsigned combo = 0; /* Gray code word: bit b set <=> row r+b is in the current sum */
  for (unsigned i = 1; i < (1u << k); ++i) {
    int b = __builtin_ctz(i);
    combo ^= 1u << b;
    const u64 *src = M + (size_t)(r + b) * ldm;
    const u64 *prev = T + (size_t)(i - 1) * ldt;
    u64 *cur = T + (size_t)i * ldt;
    for (int j = 0; j < w; ++j) cur[j] = prev[j] ^ src[j];
    L[combo] = (int)i;
  }
}

/* ---- mzd_mul_m4rm / mzd_addmul_m4rm (m4ri-sys/src/brilliantrussian.rs:210-224) ---------- */

void oracle_mul_m4rm(u64 *C, int ldc, const u64 *A, int lda, const u64 *B, int ldb, int m, int l, int n, int k,
                     int clear) {
  if (clear) zero_rows(C, ldc, m, n);
  if (m == 0 || l == 0 || n == 0) return;
  if (k == 0) k = oracle_opt_k(m, l, n);
  if (k > 16) k = 16;
  int w = width_of(n);
  u64 *T = (u64 *)malloc(((size_t)w << k) * sizeof(u64));
  int *L = (int *)malloc(sizeof(int) << k);
  /* A rows may be read one word past the last valid bit chunk: copy with a zero guard word */
  int wl = width_of(l);
  u64 *arow = (u64 *)calloc((size_t)wl + 2, sizeof(u64));
  for (int r = 0; r < l; r += k) {
    int kk = (l - r < k) ? (l - r) : k;
    oracle_make_table(B, ldb, n, r, kk, T, w, L);
    for (int i = 0; i < m; ++i) {
      memcpy(arow, A + (size_t)i * lda, (size_t)wl * sizeof(u64));
      arow[wl - 1] &= tail_mask(l);
      unsigned v = read_bits(arow, r, kk);
      const u64 *t = T + (size_t)L[v] * w;
      u64 *c = C + (size_t)i * ldc;
      for (int j = 0; j < w; ++j) c[j] ^= t[j];
    }
  }
  free(arow);
  free(L);
  free(T);
}

/* ---- mzd_mul / mzd_addmul (m4ri-sys/src/strassen.rs:8-31) ------------------------------ */

typedef struct {
  u64 *p;
  int ld;
} view;

static u64 *tmp_alloc(int rows, int cols, int *ld) {
  *ld = width_of(cols);
  return (u64 *)calloc((size_t)rows * (size_t)(*ld) + 1, sizeof(u64));
}

static void strassen_rec(u64 *C, int ldc, const u64 *A, int lda, const u64 *B, int ldb, int m, int l, int n,
                         int cutoff, int clear) {
  /* word-aligned halves; below the cutoff (or when a half would be empty) use M4RM */
  int mm = (m / 128) * 64, ll = (l / 128) * 64, nn = (n / 128) * 64;
  if (m < 2 * cutoff || l < 2 * cutoff || n < 2 * cutoff || mm == 0 || ll == 0 || nn == 0) {
    oracle_mul_m4rm(C, ldc, A, lda, B, ldb, m, l, n, 0, clear);
    return;
  }
  const int lw = ll / 64, nw = nn / 64;
  const u64 *A11 = A, *A12 = A + lw, *A21 = A + (size_t)mm * lda, *A22 = A21 + lw;
  const u64 *B11 = B, *B12 = B + nw, *B21 = B + (size_t)ll * ldb, *B22 = B21 + nw;
  u64 *C11 = C, *C12 = C + nw, *C21 = C + (size_t)mm * ldc, *C22 = C21 + nw;

  /* Winograd's variant over GF(2) (all signs are +):
   *  S1=A21+A22 S2=S1+A11 S3=A11+A21 S4=A12+S2   T1=B12+B11 T2=B22+T1 T3=B22+B12 T4=T2+B21
   *  P1=A11B11 P2=A12B21 P3=S4B22 P4=A22T4 P5=S1T1 P6=S2T2 P7=S3T3
   *  C11=P1+P2  U2=P1+P6  U3=U2+P7  U4=U2+P5  C12=U4+P3  C21=U3+P4  C22=U3+P5 */
  int lds_, ldt_, ldp;
  u64 *S1 = tmp_alloc(mm, ll, &lds_), *S2 = tmp_alloc(mm, ll, &lds_), *S3 = tmp_alloc(mm, ll, &lds_),
      *S4 = tmp_alloc(mm, ll, &lds_);
  u64 *T1 = tmp_alloc(ll, nn, &ldt_), *T2 = tmp_alloc(ll, nn, &ldt_), *T3 = tmp_alloc(ll, nn, &ldt_),
      *T4 = tmp_alloc(ll, nn, &ldt_);
  u64 *P[8];
  for (int i = 1; i <= 7; ++i) P[i] = tmp_alloc(mm, nn, &ldp);

  oracle_add(S1, lds_, A21, lda, A22, lda, mm, ll);
  oracle_add(S2, lds_, S1, lds_, A11, lda, mm, ll);
  oracle_add(S3, lds_, A11, lda, A21, lda, mm, ll);
  oracle_add(S4, lds_, A12, lda, S2, lds_, mm, ll);
  oracle_add(T1, ldt_, B12, ldb, B11, ldb, ll, nn);
  oracle_add(T2, ldt_, B22, ldb, T1, ldt_, ll, nn);
  oracle_add(T3, ldt_, B22, ldb, B12, ldb, ll, nn);
  oracle_add(T4, ldt_, T2, ldt_, B21, ldb, ll, nn);

  strassen_rec(P[1], ldp, A11, lda, B11, ldb, mm, ll, nn, cutoff, 1);
  strassen_rec(P[2], ldp, A12, lda, B21, ldb, mm, ll, nn, cutoff, 1);
  strassen_rec(P[3], ldp, S4, lds_, B22, ldb, mm, ll, nn, cutoff, 1);
  strassen_rec(P[4], ldp, A22, lda, T4, ldt_, mm, ll, nn, cutoff, 1);
  strassen_rec(P[5], ldp, S1, lds_, T1, ldt_, mm, ll, nn, cutoff, 1);
  strassen_rec(P[6], ldp, S2, lds_, T2, ldt_, mm, ll, nn, cutoff, 1);
  strassen_rec(P[7], ldp, S3, lds_, T3, ldt_, mm, ll, nn, cutoff, 1);

  /* U2 -> P6, U3 -> P7, U4 -> P6 (after U3 is formed) */
  oracle_add(P[6], ldp, P[1], ldp, P[6], ldp, mm, nn); /* U2 */
  oracle_add(P[7], ldp, P[6], ldp, P[7], ldp, mm, nn); /* U3 */
  oracle_add(P[6], ldp, P[6], ldp, P[5], ldp, mm, nn); /* U4 */
  oracle_add(P[1], ldp, P[1], ldp, P[2], ldp, mm, nn); /* C11 */
  oracle_add(P[3], ldp, P[6], ldp, P[3], ldp, mm, nn); /* C12 */
  oracle_add(P[4], ldp, P[7], ldp, P[4], ldp, mm, nn); /* C21 */
  oracle_add(P[5], ldp, P[7], ldp, P[5], ldp, mm, nn); /* C22 */

  u64 *Q[4] = {P[1], P[3], P[4], P[5]};
  u64 *D[4] = {C11, C12, C21, C22};
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < mm; ++i)
      for (int j = 0; j < nw; ++j) {
        u64 v = Q[q][(size_t)i * ldp + j];
        u64 *d = &D[q][(size_t)i * ldc + j];
        *d = clear ? v : (*d ^ v);
      }

  free(S1); free(S2); free(S3); free(S4);
  free(T1); free(T2); free(T3); free(T4);
  for (int i = 1; i <= 7; ++i) free(P[i]);

  /* peel the parts not covered by the even 2mm x 2ll x 2nn core (word-aligned offsets):
   *   C[:2mm, :2nn] += A[:2mm, 2ll:] * B[2ll:, :2nn]
   *   C[:, 2nn:]     = A * B[:, 2nn:]
   *   C[2mm:, :2nn]  = A[2mm:, :] * B[:, :2nn]                                               */
  if (l > 2 * ll)
    oracle_mul_m4rm(C, ldc, A + 2 * lw, lda, B + (size_t)2 * ll * ldb, ldb, 2 * mm, l - 2 * ll, 2 * nn, 0, 0);
  if (n > 2 * nn)
    oracle_mul_m4rm(C + 2 * nw, ldc, A, lda, B + 2 * nw, ldb, m, l, n - 2 * nn, 0, clear);
  if (m > 2 * mm)
    oracle_mul_m4rm(C + (size_t)2 * mm * ldc, ldc, A + (size_t)2 * mm * lda, lda, B, ldb, m - 2 * mm, l, 2 * nn, 0,
                    clear);
}

void oracle_mul_strassen(u64 *C, int ldc, const u64 *A, int lda, const u64 *B, int ldb, int m, int l, int n,
                         int cutoff, int clear) {
  if (cutoff <= 0) cutoff = 1024;
  if (cutoff < 64) cutoff = 64;
  /* the peeled M4RM calls rely on zero excess bits in the last column word of A views:
   * views starting at a word boundary inherit that from the parent. */
  strassen_rec(C, ldc, A, lda, B, ldb, m, l, n, cutoff, clear);
}

/* ---- tuned single-thread baseline ------------------------------------------------------ */

#define FAST_SLAB_WORDS 64 /* 4096 columns per slab: 4 tables x 256 x 64 words = 512 KiB */

static void fast_m4rm(u64 *restrict C, int ldc, const u64 *restrict A, int lda, const u64 *restrict B, int ldb, int m,
                      int l, int n, int clear) {
  const int w = width_of(n), wl = width_of(l);
  if (clear) zero_rows(C, ldc, m, n);
  if (m == 0 || l == 0 || n == 0) return;
  u64 *T = (u64 *)malloc((size_t)4 * 256 * FAST_SLAB_WORDS * sizeof(u64));
  const u64 tm = tail_mask(l);
  for (int w0 = 0; w0 < w; w0 += FAST_SLAB_WORDS) {
    const int sw = (w - w0 < FAST_SLAB_WORDS) ? (w - w0) : FAST_SLAB_WORDS;
    for (int r = 0; r < l; r += 32) {
      /* four 8-bit tables by doubling: T[v | 1<<b] = T[v] ^ B[r+8t+b] */
      for (int t = 0; t < 4; ++t) {
        u64 *Tt = T + (size_t)t * 256 * FAST_SLAB_WORDS;
        memset(Tt, 0, (size_t)sw * sizeof(u64));
        for (int b = 0; b < 8; ++b) {
          const int row = r + 8 * t + b;
          const int half = 1 << b;
          if (row < l) {
            const u64 *src = B + (size_t)row * ldb + w0;
            for (int v = 0; v < half; ++v) {
              const u64 *lo = Tt + (size_t)v * FAST_SLAB_WORDS;
              u64 *hi = Tt + (size_t)(v + half) * FAST_SLAB_WORDS;
              for (int j = 0; j < sw; ++j) hi[j] = lo[j] ^ src[j];
            }
          } else {
            for (int v = 0; v < half; ++v)
              memcpy(Tt + (size_t)(v + half) * FAST_SLAB_WORDS, Tt + (size_t)v * FAST_SLAB_WORDS,
                     (size_t)sw * sizeof(u64));
          }
        }
      }
      const int wi = r >> 6, sh = r & 63; /* r is a multiple of 32 */
      for (int i = 0; i < m; ++i) {
        u64 aw = A[(size_t)i * lda + wi];
        if (wi == wl - 1) aw &= tm;
        const unsigned v = (unsigned)(aw >> sh);
        const u64 *t0 = T + (size_t)(v & 255) * FAST_SLAB_WORDS;
        const u64 *t1 = T + (size_t)(256 + ((v >> 8) & 255)) * FAST_SLAB_WORDS;
        const u64 *t2 = T + (size_t)(512 + ((v >> 16) & 255)) * FAST_SLAB_WORDS;
        const u64 *t3 = T + (size_t)(768 + ((v >> 24) & 255)) * FAST_SLAB_WORDS;
        u64 *c = C + (size_t)i * ldc + w0;
        for (int j = 0; j < sw; ++j) c[j] ^= t0[j] ^ t1[j] ^ t2[j] ^ t3[j];
      }
    }
  }
  free(T);
}

static void fast_add(u64 *C, int ldc, const u64 *A, int lda, const u64 *B, int ldb, int rows, int w) {
  for (int i = 0; i < rows; ++i) {
    u64 *c = C + (size_t)i * ldc;
    const u64 *a = A + (size_t)i * lda, *b = B + (size_t)i * ldb;
    for (int j = 0; j < w; ++j) c[j] = a[j] ^ b[j];
  }
}

/* Strassen-Winograd on word-aligned even splits, temporaries reused as in the classic schedule */
static void fast_rec(u64 *C, int ldc, const u64 *A, int lda, const u64 *B, int ldb, int m, int l, int n) {
  const int cutoff = 4096;
  if (m < 2 * cutoff || l < 2 * cutoff || n < 2 * cutoff || (m % 128) || (l % 128) || (n % 128)) {
    fast_m4rm(C, ldc, A, lda, B, ldb, m, l, n, 1);
    return;
  }
  const int mm = m / 2, ll = l / 2, nn = n / 2, lw = ll / 64, nw = nn / 64;
  const u64 *A11 = A, *A12 = A + lw, *A21 = A + (size_t)mm * lda, *A22 = A21 + lw;
  const u64 *B11 = B, *B12 = B + nw, *B21 = B + (size_t)ll * ldb, *B22 = B21 + nw;
  u64 *C11 = C, *C12 = C + nw, *C21 = C + (size_t)mm * ldc, *C22 = C21 + nw;
  u64 *X = (u64 *)malloc((size_t)mm * lw * sizeof(u64));  /* mm x ll */
  u64 *Y = (u64 *)malloc((size_t)ll * nw * sizeof(u64));  /* ll x nn */
  u64 *P = (u64 *)malloc((size_t)mm * nw * sizeof(u64));  /* mm x nn */
  /* schedule (all + over GF(2)):
   *  X=S3=A11+A21  Y=T3=B22+B12  C21=P7=X*Y
   *  X=S1=A21+A22  Y=T1=B12+B11  C22=P5=X*Y
   *  X=S2=S1+A11   Y=T2=B22+T1   C12=P6=X*Y
   *  X=S4=A12+S2                 C11=P3=X*B22
   *  P=P1=A11*B11  C12+=P (U2)  C21+=C12 (U3)  C12+=C22 (U4)  C22+=C21 (C22=U3+P5) ... */
  fast_add(X, lw, A11, lda, A21, lda, mm, lw);
  fast_add(Y, nw, B22, ldb, B12, ldb, ll, nw);
  fast_rec(C21, ldc, X, lw, Y, nw, mm, ll, nn); /* P7 */
  fast_add(X, lw, A21, lda, A22, lda, mm, lw);
  fast_add(Y, nw, B12, ldb, B11, ldb, ll, nw);
  fast_rec(C22, ldc, X, lw, Y, nw, mm, ll, nn); /* P5 */
  fast_add(X, lw, X, lw, A11, lda, mm, lw);
  fast_add(Y, nw, B22, ldb, Y, nw, ll, nw);
  fast_rec(C12, ldc, X, lw, Y, nw, mm, ll, nn); /* P6 */
  fast_add(X, lw, A12, lda, X, lw, mm, lw);
  fast_rec(C11, ldc, X, lw, B22, ldb, mm, ll, nn); /* P3 */
  fast_rec(P, nw, A11, lda, B11, ldb, mm, ll, nn);   /* P1 */
  fast_add(C12, ldc, C12, ldc, P, nw, mm, nw);       /* U2 = P1+P6 */
  fast_add(C21, ldc, C21, ldc, C12, ldc, mm, nw);    /* U3 = U2+P7 */
  fast_add(C12, ldc, C12, ldc, C22, ldc, mm, nw);    /* U4 = U2+P5 */
  fast_add(C22, ldc, C22, ldc, C21, ldc, mm, nw);    /* C22 = U3+P5 */
  fast_add(C12, ldc, C12, ldc, C11, ldc, mm, nw);    /* C12 = U4+P3 */
  fast_add(Y, nw, Y, nw, B21, ldb, ll, nw);          /* T4 = T2+B21 */
  fast_rec(C11, ldc, A22, lda, Y, nw, mm, ll, nn);   /* P4 */
  fast_add(C21, ldc, C21, ldc, C11, ldc, mm, nw);    /* C21 = U3+P4 */
  fast_rec(C11, ldc, A12, lda, B21, ldb, mm, ll, nn); /* P2 */
  fast_add(C11, ldc, C11, ldc, P, nw, mm, nw);       /* C11 = P1+P2 */
  free(X);
  free(Y);
  free(P);
}

void oracle_mul_fast(u64 *C, int ldc, const u64 *A, int lda, const u64 *B, int ldb, int m, int l, int n) {
  fast_rec(C, ldc, A, lda, B, ldb, m, l, n);
}

/* ---- elimination -------------------------------------------------------------------------- */

int oracle_echelonize(u64 *M, int ld, int nrows, int ncols, int limit, int full, int *pivcols) {
  const int w = (ncols + 63) / 64;
  if (limit <= 0 || limit > ncols) limit = ncols;
  int r = 0;
  for (int c = 0; c < limit && r < nrows; ++c) {
    const int cw = c >> 6;
    const u64 bit = 1ull << (c & 63);
    int p = -1;
    for (int i = r; i < nrows; ++i)
      if (M[(size_t)i * ld + cw] & bit) {
        p = i;
        break;
      }
    if (p < 0) continue;
    if (p != r)
      for (int j = 0; j < w; ++j) {
        const u64 t = M[(size_t)p * ld + j];
        M[(size_t)p * ld + j] = M[(size_t)r * ld + j];
        M[(size_t)r * ld + j] = t;
      }
    const u64 *pr = M + (size_t)r * ld;
    for (int i = full ? 0 : r + 1; i < nrows; ++i)
      if (i != r && (M[(size_t)i * ld + cw] & bit)) {
        u64 *row = M + (size_t)i * ld;
        for (int j = cw; j < w; ++j) row[j] ^= pr[j]; /* the pivot row is zero left of its pivot */
      }
    if (pivcols) pivcols[r] = c;
    ++r;
  }
  return r;
}

int oracle_inverse(u64 *Ainv, int ldi, const u64 *A, int lda, int n) {
  const int nw = (n + 63) / 64, tw = 2 * nw;
  u64 *T = (u64 *)calloc((size_t)n * tw, sizeof(u64));
  for (int i = 0; i < n; ++i) {
    memcpy(T + (size_t)i * tw, A + (size_t)i * lda, (size_t)nw * sizeof(u64));
    T[(size_t)i * tw + nw + (i >> 6)] |= 1ull << (i & 63);
  }
  const int rank = oracle_echelonize(T, tw, n, nw * 64 + n, n, 1, NULL);
  if (rank == n)
    for (int i = 0; i < n; ++i) memcpy(Ainv + (size_t)i * ldi, T + (size_t)i * tw + nw, (size_t)nw * sizeof(u64));
  free(T);
  return rank == n ? 0 : -1;
}

int oracle_solve_left(const u64 *A, int lda, int m, int n, u64 *B, int ldb, int brows, int k) {
  const int nw = (n + 63) / 64, kw = (k + 63) / 64, tw = nw + kw;
  u64 *T = (u64 *)calloc((size_t)m * tw, sizeof(u64));
  int *piv = (int *)malloc(sizeof(int) * (size_t)(m < n ? m : n) + sizeof(int));
  for (int i = 0; i < m; ++i) {
    memcpy(T + (size_t)i * tw, A + (size_t)i * lda, (size_t)nw * sizeof(u64));
    memcpy(T + (size_t)i * tw + nw, B + (size_t)i * ldb, (size_t)kw * sizeof(u64));
  }
  const int rank = oracle_echelonize(T, tw, m, nw * 64 + k, n, 1, piv);
  int bad = 0;
  for (int i = rank; i < m; ++i)
    for (int j = 0; j < kw; ++j) bad |= T[(size_t)i * tw + nw + j] != 0;
  for (int i = 0; i < brows; ++i) memset(B + (size_t)i * ldb, 0, (size_t)kw * sizeof(u64));
  for (int r = 0; r < rank; ++r) memcpy(B + (size_t)piv[r] * ldb, T + (size_t)r * tw + nw, (size_t)kw * sizeof(u64));
  free(T);
  free(piv);
  return bad ? -1 : 0;
}
