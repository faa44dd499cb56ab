// gf2_small_host.cpp -- tiny products and echelon forms on the host (size dispatch of the drop-in entry points).
//
// The reference's own benchmarks multiply 10 x 10 ... 1000 x 64 x 1000 matrices (m4ri-rust/benches/binary_matrix.rs:30-76); a
// CPU does those in well under a microsecond to a few tens of microseconds, while a device call costs 30-60 us of upload,
// launch and download before the first bit is computed.  SURVEY.md section 7 step 4 plans "dispatch CPU vs GPU by size":
// below M4RI_HIP_HOST_SMALL_WORK word operations (default 2^20; 0 = everything goes to the device, which is what the GPU
// parity tests run with) mzd_mul / mzd_mul_m4rm / mzd_mul_naive / _mzd_mul_naive / _mzd_mul_va and mzd_echelonize* take the
// routines below.  This is a dispatch, not a fallback: the entry points still require a usable HIP device and fail loudly
// without one (m4ri_hip_api.cpp), and nothing here is used by bench.py or by any device-resident call.
// The code is this library's own (word-parallel Four Russians on byte-aligned chunks / row XOR by set bits / word-parallel
// Gauss-Jordan); it shares nothing with oracle/, which is test infrastructure.
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <vector>

#include "api_internal.h"

namespace {
std::atomic<long long> g_small_calls{0};

inline word masked_word(const mzd_t *M, rci_t r, wi_t j) {
  const word v = M->rows[r][j];
  return j == M->width - 1 ? (v & M->high_bitmask) : v;
}

// dst row (width w, mask = high_bitmask of the last word): whole words overwritten, bits past ncols of the last word kept
inline void merge_row(word *dst, const word *src, wi_t w, word mask) {
  for (wi_t j = 0; j + 1 < w; ++j) dst[j] = src[j];
  if (w) dst[w - 1] = (dst[w - 1] & ~mask) | (src[w - 1] & mask);
}
}  // namespace

long long gf2_small_work_limit() {
  const char *e = std::getenv("M4RI_HIP_HOST_SMALL_WORK");  // read per call: tests switch it
  return e ? std::atoll(e) : (1ll << 20);
}

bool gf2_small_product(long long m, long long l, long long n) {
  const long long lim = gf2_small_work_limit();
  if (lim <= 0) return false;
  const long long w = (n + 63) / 64;
  return m * l * w <= lim;
}

extern "C" long long gf2_host_small_calls(void) { return g_small_calls.load(); }

// C (+)= A * B, any shapes (including windows and ragged widths).  A, B, C must not alias.
extern "C" int gf2_mul_host_small(mzd_t *C, mzd_t const *A, mzd_t const *B, int accumulate) {
  if (!C || !A || !B || A->ncols != B->nrows || C->nrows != A->nrows || C->ncols != B->ncols) return -1;
  g_small_calls.fetch_add(1);
  const rci_t m = A->nrows, l = A->ncols;
  const wi_t w = C->width, wl = A->width;
  if (m == 0 || C->ncols == 0) return 0;
  std::vector<word> S((size_t)m * (w ? w : 1), 0);  // the product, dense; merged into C's rows at the end
  if (l > 0) {
    // rows of B with the excess bits of the last word cleared (windows may carry their parent's bits there)
    std::vector<word> Bd((size_t)l * w);
    for (rci_t k = 0; k < l; ++k)
      for (wi_t j = 0; j < w; ++j) Bd[(size_t)k * w + j] = masked_word(B, k, j);
    const double cost_direct = (double)m * l * 0.5 * w, cost_table = ((l + 7) / 8) * (255.0 + m) * w;
    const double cost_nt = w == 1 ? (double)l * (C->ncols * 0.5 + 1) + (double)m * C->ncols * wl : 1e300;
    if (cost_nt < cost_direct && cost_nt < cost_table) {
      // a handful of columns (matrix x vector, mul_slice: binary_matrix.rs:416-431): transpose B into one bit row per column and
      // take parities of ANDs -- what mzd_mul_naive does upstream (mzd.rs:150-168)
      const rci_t n = C->ncols;
      std::vector<word> Bt((size_t)n * wl, 0);
      for (rci_t k = 0; k < l; ++k) {
        word bits = Bd[k];
        while (bits) {
          const int j = __builtin_ctzll(bits);
          bits &= bits - 1;
          Bt[(size_t)j * wl + (k >> 6)] |= (word)1 << (k & 63);
        }
      }
      for (rci_t i = 0; i < m; ++i) {
        word out = 0;
        for (rci_t j = 0; j < n; ++j) {
          const word *bt = &Bt[(size_t)j * wl];
          word x = 0;
          for (wi_t q = 0; q < wl; ++q) x ^= masked_word(A, i, q) & bt[q];
          out |= (word)(__builtin_popcountll(x) & 1) << j;
        }
        S[i] = out;
      }
    } else if (cost_direct <= cost_table) {
      // few rows: add the rows of B selected by the set bits of each row of A
      for (rci_t i = 0; i < m; ++i) {
        word *s = &S[(size_t)i * w];
        for (wi_t q = 0; q < wl; ++q) {
          word bits = masked_word(A, i, q);
          while (bits) {
            const int b = __builtin_ctzll(bits);
            bits &= bits - 1;
            const word *br = &Bd[((size_t)q * 64 + b) * w];
            for (wi_t j = 0; j < w; ++j) s[j] ^= br[j];
          }
        }
      }
    } else {
      // Four Russians on byte-aligned chunks: T[e] = XOR of the rows of B selected by the bits of e (each entry from the
      // entry with its lowest set bit cleared: one row addition per entry), then one table row per row of A and chunk
      std::vector<word> T((size_t)256 * w);
      const int nchunks = (l + 7) / 8;
      for (int p = 0; p < nchunks; ++p) {
        const int kb = (l - 8 * p < 8) ? (l - 8 * p) : 8;  // rows of B in this chunk
        for (wi_t j = 0; j < w; ++j) T[j] = 0;
        for (int e = 1; e < (1 << kb); ++e) {
          const word *prev = &T[(size_t)(e & (e - 1)) * w], *br = &Bd[((size_t)8 * p + __builtin_ctz(e)) * w];
          word *t = &T[(size_t)e * w];
          for (wi_t j = 0; j < w; ++j) t[j] = prev[j] ^ br[j];
        }
        const wi_t q = p >> 3;
        const int sh = (p & 7) * 8;
        const unsigned sel = (1u << kb) - 1;
        for (rci_t i = 0; i < m; ++i) {
          const unsigned e = (unsigned)(masked_word(A, i, q) >> sh) & sel;
          if (!e) continue;
          const word *t = &T[(size_t)e * w];
          word *s = &S[(size_t)i * w];
          for (wi_t j = 0; j < w; ++j) s[j] ^= t[j];
        }
      }
    }
  }
  for (rci_t i = 0; i < m; ++i) {
    word *s = &S[(size_t)i * w];
    if (accumulate)
      for (wi_t j = 0; j < w; ++j) s[j] ^= C->rows[i][j];
    merge_row(C->rows[i], s, w, C->high_bitmask);
  }
  return 0;
}

// C (+)= A * Bt^T with Bt given transposed (mzd.rs:154-168): bit (i, j) = parity of the AND of row i of A and row j of Bt
extern "C" int gf2_mul_nt_host_small(mzd_t *C, mzd_t const *A, mzd_t const *Bt, int accumulate) {
  if (!C || !A || !Bt || A->ncols != Bt->ncols || C->nrows != A->nrows || C->ncols != Bt->nrows) return -1;
  g_small_calls.fetch_add(1);
  const rci_t m = A->nrows, n = Bt->nrows;
  const wi_t wl = A->width, w = C->width;
  std::vector<word> s(w ? w : 1);
  for (rci_t i = 0; i < m; ++i) {
    for (wi_t j = 0; j < w; ++j) s[j] = accumulate ? C->rows[i][j] : 0;
    for (rci_t j = 0; j < n; ++j) {
      word x = 0;
      for (wi_t q = 0; q < wl; ++q) x ^= masked_word(A, i, q) & masked_word(Bt, j, q);
      s[j >> 6] ^= (word)(__builtin_popcountll(x) & 1) << (j & 63);
    }
    merge_row(C->rows[i], s.data(), w, C->high_bitmask);
  }
  return 0;
}

// In-place (reduced if full != 0) row echelon form by word-parallel Gauss-Jordan; returns the rank.  full == 0 clears below
// the pivots only (the row contents of a non-reduced form are not contractual: rank, pivot columns, shape and row space are).
extern "C" int gf2_echelonize_host_small(mzd_t *A, int full) {
  g_small_calls.fetch_add(1);
  const rci_t m = A->nrows, n = A->ncols;
  const wi_t w = A->width;
  if (m == 0 || n == 0) return 0;
  std::vector<word> M((size_t)m * w);
  for (rci_t i = 0; i < m; ++i)
    for (wi_t j = 0; j < w; ++j) M[(size_t)i * w + j] = masked_word(A, i, j);
  rci_t rank = 0;
  for (rci_t c = 0; c < n && rank < m; ++c) {
    const wi_t cw = c >> 6;
    const word bit = (word)1 << (c & 63);
    rci_t p = rank;
    while (p < m && !(M[(size_t)p * w + cw] & bit)) ++p;
    if (p == m) continue;
    if (p != rank)
      for (wi_t j = cw; j < w; ++j) std::swap(M[(size_t)p * w + j], M[(size_t)rank * w + j]);
    const word *pr = &M[(size_t)rank * w];
    for (rci_t i = full ? 0 : rank + 1; i < m; ++i) {
      if (i == rank || !(M[(size_t)i * w + cw] & bit)) continue;
      word *r = &M[(size_t)i * w];
      for (wi_t j = cw; j < w; ++j) r[j] ^= pr[j];
    }
    ++rank;
  }
  for (rci_t i = 0; i < m; ++i) merge_row(A->rows[i], &M[(size_t)i * w], w, A->high_bitmask);
  return rank;
}
