// mzd_host.cpp -- host-side mzd_t container of libm4ri_hip.so.
//
// Keeps M4RI's 64-byte mzd_t byte for byte (m4ri-sys/src/mzd.rs:24-79): the Rust wrapper
// dereferences rows / nrows / ncols directly (m4ri-rust/src/friendly/binary_matrix.rs:138,176,286)
// and recomputes row addresses from blocks / offset_vector / rowstride (mzd.rs:277-313).
// Matrices are always single-block; blocks of >= pin threshold are allocated as pinned host
// memory so that the multiply entry points can DMA them at PCIe rate.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <emmintrin.h>
#include <map>
#include <mutex>

#include "../../include/m4ri_hip.h"
#include "api_internal.h"

static_assert(sizeof(mzd_t) == 64, "mzd_t must stay 64 bytes (mzd.rs:385)");
static_assert(offsetof(mzd_t, flags) == 24 && offsetof(mzd_t, high_bitmask) == 40 && offsetof(mzd_t, blocks) == 48 &&
                  offsetof(mzd_t, rows) == 56,
              "mzd_t field offsets are part of the ABI");

static const wi_t mzd_paddingwidth = 3;
enum { kAllocMalloc = 0, kAllocPinned = 1, kAllocInline = 2 };  // inline: header, block record, rows and row pointers in ONE allocation

[[noreturn]] void gf2_die(const char *msg) {
  // M4RI's m4ri_die: print and abort (dimension mismatches are programming errors upstream too)
  std::fprintf(stderr, "m4ri_hip: %s\n", msg);
  std::abort();
}

static size_t pin_threshold() {
  static size_t t = [] {
    const char *e = std::getenv("M4RI_HIP_PIN_MIN_BYTES");
    return e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)1 << 20;
  }();
  return t;
}

// Pinned blocks are expensive to create (hipHostMalloc pins pages: ~100 ms for 512 MiB), so freed ones are kept in a
// small size-keyed pool and handed out again (M4RI keeps a similar cache of its own blocks, m4ri_mmc).
// A pooled block keeps its ROW-POINTER ARRAY: rows[i] = begin + i * rowstride is a function of the block's address and the
// shape alone, and for a tall thin matrix the array is as large as the data (2^20 x 1, the result of every `&A * &v` of an
// LPN solver, binary_matrix.rs:416-431: 8 MiB of pointers beside 8 MiB of words -- allocating and filling it took 226 us of a
// 565-us call, profiles/r05_av_breakdown.txt).  The next matrix of the same shape takes block and array as they are.
namespace {
struct PinEntry {
  void *p;
  word **rows;  // may be null (blocks that never carried a matrix: gf2_pinned_alloc)
  rci_t nrows;
  wi_t rowstride;
};
std::mutex g_pin_mu;
std::multimap<size_t, PinEntry> g_pin_free;
size_t g_pin_cached = 0;
size_t pin_cache_limit() {
  static size_t t = [] {
    const char *e = std::getenv("M4RI_HIP_PIN_CACHE_BYTES");
    return e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)8 << 30;
  }();
  return t;
}

word **make_rows(word *begin, rci_t r, wi_t rowstride) {
  word **rows = static_cast<word **>(std::malloc(((size_t)r + 1) * sizeof(word *)));
  if (!rows) gf2_die("out of memory");
  for (rci_t i = 0; i < r; ++i) rows[i] = begin + (size_t)i * rowstride;
  rows[r] = nullptr;
  return rows;
}

// a pooled pinned block of exactly `bytes`, preferring one whose row-pointer array fits (r, rowstride); *rows is that array or null
void *pin_pool_take(size_t bytes, rci_t r, wi_t rowstride, word ***rows) {
  PinEntry e{nullptr, nullptr, 0, 0};
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    auto range = g_pin_free.equal_range(bytes);
    auto pick = range.first;
    for (auto it = range.first; it != range.second; ++it)
      if (it->second.rows && it->second.nrows == r && it->second.rowstride == rowstride) {
        pick = it;
        break;
      }
    if (pick == range.second) return nullptr;
    e = pick->second;
    g_pin_cached -= bytes;
    g_pin_free.erase(pick);
  }
  if (e.rows && (e.nrows != r || e.rowstride != rowstride || !rows)) {
    std::free(e.rows);
    e.rows = nullptr;
  }
  if (rows) *rows = e.rows;
  return e.p;
}

bool pin_pool_give(void *p, size_t bytes, word **rows, rci_t r, wi_t rowstride) {
  std::lock_guard<std::mutex> lk(g_pin_mu);
  if (g_pin_cached + bytes > pin_cache_limit()) return false;
  g_pin_free.emplace(bytes, PinEntry{p, rows, r, rowstride});
  g_pin_cached += bytes;
  return true;
}
}  // namespace

// *rows: in = null; out = a row-pointer array for (r, rowstride) that came with a pooled block, or still null
static void *block_alloc(size_t bytes, uint8_t *kind, bool zero, rci_t r = 0, wi_t rowstride = 0, word ***rows = nullptr) {
  *kind = kAllocMalloc;
  if (bytes >= pin_threshold() && gf2_device_count() > 0) {
    void *p = pin_pool_take(bytes, r, rowstride, rows);
    if (!p && hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      p = nullptr;
    }
    if (p) {
      if (zero) std::memset(p, 0, bytes);
      *kind = kAllocPinned;
      return p;
    }
  }
  void *p = nullptr;
  if (posix_memalign(&p, 64, bytes ? bytes : 64) != 0) gf2_die("out of memory");
  if (zero) std::memset(p, 0, bytes);
  return p;
}

// `rows` (may be null) is the block's row-pointer array: it stays with a pooled block and is freed otherwise
static void block_free(void *p, uint8_t kind, size_t bytes, word **rows = nullptr, rci_t r = 0, wi_t rowstride = 0) {
  if (p && kind == kAllocPinned) {
    if (pin_pool_give(p, bytes, rows, r, rowstride)) return;
    (void)hipHostFree(p);
  } else if (p) {
    std::free(p);
  }
  std::free(rows);
}

// Small pinned scratch blocks for the rest of the library (the packed side copy of a thin product, m4ri_hip_api.cpp): same pool,
// whatever the size; null without a device or when pinning fails.
void *gf2_pinned_alloc(size_t bytes) {
  if (!bytes || gf2_device_count() <= 0) return nullptr;
  void *p = pin_pool_take(bytes, 0, 0, nullptr);
  if (!p && hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    p = nullptr;
  }
  return p;
}
// is the block of M pinned host memory (device-visible: a kernel may read or write it directly)?
bool gf2_mzd_block_is_pinned(mzd_t const *M) { return M && M->blocks && M->padding[0] == kAllocPinned; }
void gf2_pinned_free(void *p, size_t bytes) {
  if (p && !pin_pool_give(p, bytes, nullptr, 0, 0)) (void)hipHostFree(p);
}

// Pre-creates `count` pinned blocks of the size an r x c matrix takes and puts them into the pool: the first mzd_init /
// product / transposition of that size then finds its block instead of pinning fresh pages (hipHostMalloc: ~100 ms for the
// 512 MiB of a 65536^2 matrix -- the whole of a first `transposed()` call, 121 ms against 20 with a pooled block).
// Returns the number of blocks added (0 without a device, below the pinning threshold or beyond M4RI_HIP_PIN_CACHE_BYTES).
extern "C" int gf2_mzd_prewarm(rci_t r, rci_t c, int count) {
  if (r <= 0 || c <= 0 || count <= 0 || gf2_device_count() <= 0) return 0;
  const wi_t width = (c + m4ri_radix - 1) / m4ri_radix;
  const wi_t rowstride = (width < mzd_paddingwidth || (width & 1) == 0) ? width : width + 1;
  const size_t bytes = (size_t)r * (size_t)rowstride * sizeof(word);
  if (bytes < pin_threshold()) return 0;
  int added = 0;
  for (int i = 0; i < count; ++i) {
    {
      std::lock_guard<std::mutex> lk(g_pin_mu);
      if (g_pin_cached + bytes > pin_cache_limit()) break;
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      break;
    }
    word **rows = make_rows(static_cast<word *>(p), r, rowstride);  // the block arrives with its row pointers
    if (!pin_pool_give(p, bytes, rows, r, rowstride)) {
      std::free(rows);
      (void)hipHostFree(p);
      break;
    }
    ++added;
  }
  return added;
}

static mzd_t *mzd_init_impl(rci_t r, rci_t c, bool zero);
extern "C" mzd_t *mzd_init(rci_t r, rci_t c) { return mzd_init_impl(r, c, true); }
// product destinations are overwritten entirely: no need to clear half a gigabyte first
mzd_t *gf2_mzd_init_uncleared(rci_t r, rci_t c) { return mzd_init_impl(r, c, false); }

static mzd_t *mzd_init_impl(rci_t r, rci_t c, bool zero) {
  if (r < 0 || c < 0) gf2_die("mzd_init: negative dimension");
  const wi_t width = (c + m4ri_radix - 1) / m4ri_radix;
  const wi_t rowstride = (width < mzd_paddingwidth || (width & 1) == 0) ? width : width + 1;
  const size_t bytes = (size_t)r * (size_t)rowstride * sizeof(word);
  // Small matrices (the reference's vector <-> matrix conversions create and drop several 1 x n and n x 1 matrices per product,
  // binary_matrix.rs:416-431,332-361) live in ONE allocation: header | block record | rows | row pointers.
  const bool inl = r && c && bytes <= (size_t)64 * 1024;
  const size_t data_off = 192, rows_off = inl ? data_off + ((bytes + 63) & ~(size_t)63) : 0;
  const size_t total = inl ? rows_off + ((size_t)r + 1) * sizeof(word *) : sizeof(mzd_t);
  mzd_t *A = nullptr;
  if (posix_memalign(reinterpret_cast<void **>(&A), 64, total) != 0) gf2_die("out of memory");
  std::memset(A, 0, sizeof(mzd_t));
  A->nrows = r;
  A->ncols = c;
  A->width = width;
  A->rowstride = rowstride;
  A->high_bitmask = (c % m4ri_radix) ? ((m4ri_one << (c % m4ri_radix)) - 1) : m4ri_ffff;
  A->flags = (A->high_bitmask != m4ri_ffff) ? mzd_flag_nonzero_excess : 0;
  A->offset_vector = 0;
  A->row_offset = 0;
  uint8_t lg = 0;
  while (((long long)1 << lg) < (long long)(r > 1 ? r : 1)) ++lg;
  A->blockrows_log = lg;  // single block: (row_offset + row) >> blockrows_log == 0 for every row
  if (r && c) {
    mzd_block_t *blocks;
    uint8_t kind;
    if (inl) {
      unsigned char *base = reinterpret_cast<unsigned char *>(A);
      blocks = reinterpret_cast<mzd_block_t *>(base + 64);
      std::memset(blocks, 0, 2 * sizeof(mzd_block_t));  // [1] = terminator
      blocks[0].begin = reinterpret_cast<word *>(base + data_off);
      if (zero) std::memset(blocks[0].begin, 0, bytes);
      kind = kAllocInline;
      A->rows = reinterpret_cast<word **>(base + rows_off);
    } else {
      blocks = static_cast<mzd_block_t *>(std::calloc(2, sizeof(mzd_block_t)));  // [1] = terminator
      if (!blocks) gf2_die("out of memory");
      word **pooled_rows = nullptr;
      blocks[0].begin = static_cast<word *>(block_alloc(bytes, &kind, zero, r, rowstride, &pooled_rows));
      A->rows = pooled_rows ? pooled_rows : make_rows(blocks[0].begin, r, rowstride);
    }
    blocks[0].size = bytes;
    blocks[0].end = blocks[0].begin + (size_t)r * A->rowstride;
    A->padding[0] = kind;
    A->blocks = blocks;
    if (inl) {
      for (rci_t i = 0; i < r; ++i) A->rows[i] = blocks[0].begin + (size_t)i * A->rowstride;
      A->rows[r] = nullptr;
    }
  }
  return A;
}

static bool is_windowed(const mzd_t *A) { return (A->flags & mzd_flag_windowed_zerooffset) != 0; }
static bool owns_blocks(const mzd_t *A) {
  return A->blocks && (!is_windowed(A) || (A->flags & mzd_flag_windowed_ownsblocks));
}

extern "C" void mzd_free(mzd_t *A) {
  if (!A) return;
  // the operand cache is keyed by the block: it goes with the block's owner (a later block may get the same address);
  // freeing a window of a cached parent leaves the parent's device copy alone
  if (owns_blocks(A) || !A->blocks) gf2_cache_forget(A);
  if (owns_blocks(A) && A->padding[0] == kAllocInline) {  // one allocation holds everything
    std::free(A);
    return;
  }
  if (owns_blocks(A)) {
    block_free(A->blocks[0].begin, A->padding[0], A->blocks[0].size, A->rows, A->nrows, A->rowstride);  // takes the row pointers along
    std::free(A->blocks);
  } else {
    std::free(A->rows);
  }
  std::free(A);
}

extern "C" mzd_t *mzd_init_window(mzd_t *M, rci_t lowr, rci_t lowc, rci_t highr, rci_t highc) {
  if (lowc % m4ri_radix) gf2_die("mzd_init_window: lowc must be a multiple of 64");
  if (lowr < 0 || lowc < 0 || highr > M->nrows || highc > M->ncols || highr < lowr || highc < lowc)
    gf2_die("mzd_init_window: window out of range");
  mzd_t *W = nullptr;
  if (posix_memalign(reinterpret_cast<void **>(&W), 64, sizeof(mzd_t)) != 0) gf2_die("out of memory");
  std::memset(W, 0, sizeof(mzd_t));
  W->nrows = highr - lowr;
  W->ncols = highc - lowc;
  W->width = (W->ncols + m4ri_radix - 1) / m4ri_radix;
  W->rowstride = M->rowstride;
  W->high_bitmask = (W->ncols % m4ri_radix) ? ((m4ri_one << (W->ncols % m4ri_radix)) - 1) : m4ri_ffff;
  W->flags = mzd_flag_windowed_zerooffset;
  W->flags |= (W->ncols % m4ri_radix == 0) ? mzd_flag_windowed_zeroexcess : mzd_flag_nonzero_excess;
  W->blockrows_log = M->blockrows_log;
  W->row_offset = M->row_offset + lowr;
  W->offset_vector = M->offset_vector + lowr * M->rowstride + lowc / m4ri_radix;
  W->blocks = M->blocks;
  W->padding[0] = M->padding[0];
  if (W->nrows) {
    W->rows = static_cast<word **>(std::malloc(((size_t)W->nrows + 1) * sizeof(word *)));
    if (!W->rows) gf2_die("out of memory");
    for (rci_t i = 0; i < W->nrows; ++i) W->rows[i] = M->rows[lowr + i] + lowc / m4ri_radix;
    W->rows[W->nrows] = nullptr;
  }
  return W;
}

static inline void copy_row_masked(word *d, const word *s, wi_t width, word mask) {
  if (width <= 0) return;
  if (width > 1) std::memcpy(d, s, (size_t)(width - 1) * sizeof(word));
  d[width - 1] = (d[width - 1] & ~mask) | (s[width - 1] & mask);
}

extern "C" mzd_t *mzd_copy(mzd_t *N, mzd_t const *P) {
  if (N) gf2_cache_forget(N);  // written below: a device copy kept for it is stale
  if (N == P) return N;
  if (!N)
    N = mzd_init(P->nrows, P->ncols);
  else if (N->nrows < P->nrows || N->ncols < P->ncols)
    gf2_die("mzd_copy: Target matrix is too small.");
  for (rci_t i = 0; i < P->nrows; ++i) copy_row_masked(N->rows[i], P->rows[i], P->width, P->high_bitmask);
  return N;
}

extern "C" int mzd_equal(mzd_t const *A, mzd_t const *B) {
  if (A->nrows != B->nrows || A->ncols != B->ncols) return 0;
  if (A == B) return 1;
  const wi_t w = A->width;
  for (rci_t i = 0; i < A->nrows; ++i) {
    const word *a = A->rows[i], *b = B->rows[i];
    for (wi_t j = 0; j + 1 < w; ++j)
      if (a[j] != b[j]) return 0;
    if (w && ((a[w - 1] ^ b[w - 1]) & A->high_bitmask)) return 0;
  }
  return 1;
}

extern "C" int mzd_is_zero(mzd_t const *A) {
  for (rci_t i = 0; i < A->nrows; ++i) {
    const word *a = A->rows[i];
    for (wi_t j = 0; j + 1 < A->width; ++j)
      if (a[j]) return 0;
    if (A->width && (a[A->width - 1] & A->high_bitmask)) return 0;
  }
  return 1;
}

extern "C" void mzd_randomize(mzd_t *A) {
  if (A) gf2_cache_forget(A);  // written below: a device copy kept for it is stale
  // M4RI draws from libc random(); here a process-wide counter-based splitmix64 stream (thread safe,
  // successive calls give fresh bits as the reference's tests expect: mzd.rs:389-394).
  static std::atomic<uint64_t> ctr{0x243F6A8885A308D3ull};
  const uint64_t nwords = (uint64_t)A->nrows * (uint64_t)A->width;
  uint64_t t = ctr.fetch_add(nwords + 1);
  for (rci_t i = 0; i < A->nrows; ++i) {
    word *a = A->rows[i];
    for (wi_t j = 0; j < A->width; ++j) {
      uint64_t z = (t += 0x9E3779B97F4A7C15ull);
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      if (j == A->width - 1)
        a[j] = (a[j] & ~A->high_bitmask) | (z & A->high_bitmask);
      else
        a[j] = z;
    }
  }
}

extern "C" void mzd_set_ui(mzd_t *A, unsigned int value) {
  if (A) gf2_cache_forget(A);  // written below: a device copy kept for it is stale
  for (rci_t i = 0; i < A->nrows; ++i) {
    word *a = A->rows[i];
    for (wi_t j = 0; j + 1 < A->width; ++j) a[j] = 0;
    if (A->width) a[A->width - 1] &= ~A->high_bitmask;
  }
  if (value % 2 == 0) return;
  const rci_t k = A->nrows < A->ncols ? A->nrows : A->ncols;
  for (rci_t i = 0; i < k; ++i) A->rows[i][i / m4ri_radix] |= m4ri_one << (i % m4ri_radix);
}

// 64x64 bit-block transpose (recursive block swap), in place on 64 words
static inline void transpose64(word x[64]) {
  word mask = 0x00000000FFFFFFFFull;
  for (int d = 32; d >= 1; d >>= 1, mask ^= mask << d) {
    for (int k = 0; k < 64; k = (k + d + 1) & ~d) {
      const word t = ((x[k] >> d) ^ x[k + d]) & mask;
      x[k] ^= t << d;
      x[k + d] ^= t;
    }
  }
}

extern "C" mzd_t *mzd_transpose(mzd_t *DST, mzd_t const *A) {
  if (DST) gf2_cache_forget(DST);  // written below: a device copy kept for it is stale
  if (DST && (DST->nrows != A->ncols || DST->ncols != A->nrows)) gf2_die("mzd_transpose: Wrong size for return matrix.");
  // a fresh thin product carries its transposed form already (as_vector of every `&A * &v`, binary_matrix.rs:332-361)
  if (A->ncols <= 64 && A->nrows >= 64 && DST != A)
    if (mzd_t *D = gf2_transpose_from_side_copy(DST, A)) return D;
  // large matrices: PCIe both ways plus the device kernel is ~100x faster than the host loop below
  static const long long gpu_min_bits = [] {
    const char *e = std::getenv("M4RI_HIP_TRANSPOSE_GPU_MIN_BITS");
    return e ? std::atoll(e) : (1ll << 24);
  }();
  const bool big = A->nrows > 0 && A->ncols > 0 && (long long)A->nrows * A->ncols >= gpu_min_bits && gf2_device_count() > 0;
  if (big && !(DST && (DST->flags & mzd_flag_windowed_zerooffset))) {
    mzd_t *D = DST ? DST : gf2_mzd_init_uncleared(A->ncols, A->nrows);  // every word is overwritten by the download
    if (gf2_host_transpose_gpu(D, A) == 0) return D;
    if (!DST) mzd_free(D);
  }
  if (!DST) DST = mzd_init(A->ncols, A->nrows);
  if (A->nrows == 0 || A->ncols == 0) return DST;
  // column <-> row vectors: what the friendly layer's vector products do twice per call (binary_matrix.rs:426,335)
  if (A->ncols == 1) {  // n x 1 -> 1 x n: bit 0 of every row (as_vector of every `&A * &v`: 2^20 rows in the LPN configuration)
    word *d = DST->rows[0];
    const bool dense = A->rowstride == 1 && !(A->flags & mzd_flag_windowed_zerooffset);  // one word per row, rows back to back
    for (wi_t j = 0; j < DST->width; ++j) {
      word v = 0;
      const rci_t lim = (A->nrows - 64 * j < 64) ? (A->nrows - 64 * j) : 64;
      if (dense && lim == 64) {  // two rows per step: bit 0 moved to the sign position, gathered by movmskpd (SSE2: x86-64 baseline)
        const word *src = A->rows[64 * j];
        for (int r = 0; r < 64; r += 2) {
          const __m128i x = _mm_slli_epi64(_mm_loadu_si128(reinterpret_cast<const __m128i *>(src + r)), 63);
          v |= (word)_mm_movemask_pd(_mm_castsi128_pd(x)) << r;
        }
      } else {
        for (rci_t r = 0; r < lim; ++r) v |= (A->rows[64 * j + r][0] & 1) << r;
      }
      d[j] = (j == DST->width - 1) ? ((d[j] & ~DST->high_bitmask) | (v & DST->high_bitmask)) : v;
    }
    return DST;
  }
  if (A->nrows == 1) {  // 1 x n -> n x 1
    const word *a = A->rows[0];
    for (rci_t c = 0; c < A->ncols; ++c) {
      word *d = DST->rows[c];
      *d = (*d & ~DST->high_bitmask) | ((a[c >> 6] >> (c & 63)) & 1);
    }
    return DST;
  }
  word blk[64];
  for (rci_t bi = 0; bi < A->nrows; bi += 64) {
    const int nr = (A->nrows - bi < 64) ? (A->nrows - bi) : 64;
    for (wi_t bj = 0; bj < A->width; ++bj) {
      for (int r = 0; r < 64; ++r) {
        word v = (r < nr) ? A->rows[bi + r][bj] : 0;
        if (bj == A->width - 1) v &= A->high_bitmask;
        blk[r] = v;
      }
      transpose64(blk);
      const int nc = (A->ncols - 64 * bj < 64) ? (A->ncols - 64 * bj) : 64;
      const wi_t dw = bi / 64;
      for (int c = 0; c < nc; ++c) {
        word *d = DST->rows[64 * bj + c] + dw;
        if (dw == DST->width - 1)
          *d = (*d & ~DST->high_bitmask) | (blk[c] & DST->high_bitmask);
        else
          *d = blk[c];
      }
    }
  }
  return DST;
}

extern "C" mzd_t *mzd_add(mzd_t *C, mzd_t const *A, mzd_t const *B) {
  if (C) gf2_cache_forget(C);  // written below: a device copy kept for it is stale
  if (A->nrows != B->nrows || A->ncols != B->ncols) gf2_die("mzd_add: rows and columns must match.");
  if (!C)
    C = mzd_init(A->nrows, A->ncols);
  else if (C != A && (C->nrows != A->nrows || C->ncols != A->ncols))
    gf2_die("mzd_add: rows and columns of returned matrix must match.");
  const wi_t w = A->width;
  for (rci_t i = 0; i < A->nrows; ++i) {
    word *c = C->rows[i];
    const word *a = A->rows[i], *b = B->rows[i];
    for (wi_t j = 0; j + 1 < w; ++j) c[j] = a[j] ^ b[j];
    if (w) c[w - 1] = (c[w - 1] & ~C->high_bitmask) | ((a[w - 1] ^ b[w - 1]) & C->high_bitmask);
  }
  return C;
}

extern "C" mzd_t *mzd_sub(mzd_t *C, mzd_t const *A, mzd_t const *B) { return mzd_add(C, A, B); }

static inline BIT read_bit(const mzd_t *M, rci_t r, rci_t c) {
  return (BIT)((M->rows[r][c / m4ri_radix] >> (c % m4ri_radix)) & 1);
}
static inline void write_bit(mzd_t *M, rci_t r, rci_t c, BIT v) {
  word *w = &M->rows[r][c / m4ri_radix];
  *w = (*w & ~(m4ri_one << (c % m4ri_radix))) | ((word)(v & 1) << (c % m4ri_radix));
}

extern "C" mzd_t *mzd_concat(mzd_t *C, mzd_t const *A, mzd_t const *B) {
  if (C) gf2_cache_forget(C);  // written below: a device copy kept for it is stale
  if (A->nrows != B->nrows) gf2_die("mzd_concat: Bad arguments to concat!");
  if (!C)
    C = mzd_init(A->nrows, A->ncols + B->ncols);
  else if (C->nrows != A->nrows || C->ncols != A->ncols + B->ncols)
    gf2_die("mzd_concat: C has wrong dimension!");
  for (rci_t i = 0; i < A->nrows; ++i) {
    copy_row_masked(C->rows[i], A->rows[i], A->width, A->high_bitmask);
    for (rci_t j = 0; j < B->ncols; ++j) write_bit(C, i, A->ncols + j, read_bit(B, i, j));
  }
  return C;
}

extern "C" mzd_t *mzd_stack(mzd_t *C, mzd_t const *A, mzd_t const *B) {
  if (C) gf2_cache_forget(C);  // written below: a device copy kept for it is stale
  if (A->ncols != B->ncols) gf2_die("mzd_stack: A->ncols != B->ncols!");
  if (!C)
    C = mzd_init(A->nrows + B->nrows, A->ncols);
  else if (C->nrows != A->nrows + B->nrows || C->ncols != A->ncols)
    gf2_die("mzd_stack: C has wrong dimension!");
  for (rci_t i = 0; i < A->nrows; ++i) copy_row_masked(C->rows[i], A->rows[i], A->width, A->high_bitmask);
  for (rci_t i = 0; i < B->nrows; ++i) copy_row_masked(C->rows[A->nrows + i], B->rows[i], B->width, B->high_bitmask);
  return C;
}

extern "C" mzd_t *mzd_submatrix(mzd_t *S, mzd_t const *M, rci_t lowr, rci_t lowc, rci_t highr, rci_t highc) {
  if (S) gf2_cache_forget(S);  // written below: a device copy kept for it is stale
  const rci_t nrows = highr - lowr, ncols = highc - lowc;
  if (!S)
    S = mzd_init(nrows, ncols);
  else if (S->nrows < nrows || S->ncols < ncols)
    gf2_die("mzd_submatrix: got S with wrong dimensions");
  for (rci_t i = 0; i < nrows; ++i)
    for (rci_t j = 0; j < ncols; ++j) write_bit(S, i, j, read_bit(M, lowr + i, lowc + j));
  return S;
}

extern "C" void mzd_row_swap(mzd_t *M, rci_t a, rci_t b) {
  if (M) gf2_cache_forget(M);  // written below: a device copy kept for it is stale
  if (a == b) return;
  word *x = M->rows[a], *y = M->rows[b];
  for (wi_t j = 0; j < M->width; ++j) {
    const word mask = (j == M->width - 1) ? M->high_bitmask : m4ri_ffff;
    const word t = (x[j] ^ y[j]) & mask;
    x[j] ^= t;
    y[j] ^= t;
  }
}

extern "C" void mzd_copy_row(mzd_t *B, rci_t i, mzd_t const *A, rci_t j) {
  if (B) gf2_cache_forget(B);  // written below: a device copy kept for it is stale
  if (A->ncols > B->ncols) gf2_die("mzd_copy_row: source wider than target");
  copy_row_masked(B->rows[i], A->rows[j], A->width, A->high_bitmask);
}

// mzd_col_swap (m4ri-sys/src/mzd.rs:144): swap two columns in every row.
extern "C" void mzd_col_swap(mzd_t *M, rci_t cola, rci_t colb) {
  if (M) gf2_cache_forget(M);  // written below: a device copy kept for it is stale
  if (cola == colb) return;
  if (cola < 0 || colb < 0 || cola >= M->ncols || colb >= M->ncols) gf2_die("mzd_col_swap: column out of range");
  const wi_t wa = cola / m4ri_radix, wb = colb / m4ri_radix;
  const int sa = cola % m4ri_radix, sb = colb % m4ri_radix;
  for (rci_t i = 0; i < M->nrows; ++i) {
    word *row = M->rows[i];
    const word d = ((row[wa] >> sa) ^ (row[wb] >> sb)) & m4ri_one;  // 1 where the two bits differ
    row[wa] ^= d << sa;
    row[wb] ^= d << sb;
  }
}

// mzd_row_clear_offset (m4ri-sys/src/mzd.rs:235-240): clear row `row` from column `coloffset` to its end.
extern "C" void mzd_row_clear_offset(mzd_t *M, rci_t row, rci_t coloffset) {
  if (M) gf2_cache_forget(M);
  if (row < 0 || row >= M->nrows || coloffset < 0) gf2_die("mzd_row_clear_offset: out of range");
  if (coloffset >= M->ncols) return;
  const wi_t w0 = coloffset / m4ri_radix;
  const int s = coloffset % m4ri_radix;
  word *r = M->rows[row];
  const word keep = s ? ((m4ri_one << s) - 1) : 0;  // bits below the offset inside its word
  if (w0 == M->width - 1) {
    r[w0] &= keep | ~M->high_bitmask;  // a window's foreign bits past its width stay
    return;
  }
  r[w0] &= keep;
  for (wi_t j = w0 + 1; j < M->width - 1; ++j) r[j] = 0;
  r[M->width - 1] &= ~M->high_bitmask;
}

// mzd_invert_naive (m4ri-sys/src/mzd.rs:214-218): inverse by Gaussian elimination of [A | I]; `identity` may be passed in to
// save building it, `inv` may be NULL (allocated).  The elimination is this library's own mzd_echelonize (device, or the
// host routine of the size dispatch); returns NULL for a singular matrix.
extern "C" mzd_t *mzd_invert_naive(mzd_t *inv, mzd_t const *A, mzd_t const *identity) {
  if (!A || A->nrows != A->ncols) gf2_die("mzd_invert_naive: matrix must be square");
  const rci_t n = A->nrows;
  if (identity && (identity->nrows != n || identity->ncols != n)) gf2_die("mzd_invert_naive: identity has wrong dimensions");
  if (inv && (inv->nrows != n || inv->ncols != n)) gf2_die("mzd_invert_naive: inv has wrong dimensions");
  if (n == 0) return inv ? inv : mzd_init(0, 0);
  mzd_t *I = nullptr;
  if (!identity) {
    I = mzd_init(n, n);
    mzd_set_ui(I, 1);
  }
  mzd_t *H = mzd_concat(nullptr, A, identity ? identity : I);
  if (I) mzd_free(I);
  // [A | I] always has rank n (the identity block), so the rank says nothing about A.  What does: in the FULLY REDUCED echelon
  // form (full = 1 -- both the device path and the small-size host path of mzd_echelonize deliver it; tests/test_gpu_elim.py::
  // test_invert_naive runs both) the left block is the identity exactly when A is invertible -- a singular A leaves a pivot
  // right of column n and a zero on the diagonal.  (Upstream M4RI returns NULL for a singular A only when the rank is 0; here
  // every singular A gives NULL: INTEGRATION.md section 3.)
  (void)mzd_echelonize(H, 1);
  bool ok = true;
  for (rci_t i = 0; ok && i < n; ++i) ok = read_bit(H, i, i) == 1;
  mzd_t *out = nullptr;
  if (ok) out = mzd_submatrix(inv, H, 0, n, n, 2 * n);
  mzd_free(H);
  return out;
}

// mzd_make_table (brilliantrussian.rs:8-17): T (2^k rows, preallocated, M->ncols columns) receives every XOR combination of the k
// rows r .. r + k - 1 of M, L (2^k entries) the row of T that holds the combination selected by a k-bit value v (bit j of v <->
// row r + j): T[L[v]] = XOR of the rows r + j with bit j of v set.  Rows of T are generated in Gray-code order (each from its
// predecessor by ONE row addition), as the declaration says; only the words from column c on are written, like upstream
// (the elimination routines that call it never look left of their current column).  The device kernels build their tables
// in LDS themselves (gf2_kernels.hip); this host form completes the declared surface of section 8 row a9.
extern "C" void mzd_make_table(mzd_t const *M, rci_t r, rci_t c, int k, mzd_t *T, rci_t *L) {
  if (k < 0 || k > 24 || r < 0 || c < 0 || r + k > M->nrows || T->nrows < (1 << k) || T->ncols < M->ncols)
    gf2_die("mzd_make_table: bad arguments");
  gf2_cache_forget(T);
  const wi_t w0 = c / m4ri_radix, w = M->width;
  for (wi_t j = w0; j < w; ++j) T->rows[0][j] = 0;
  L[0] = 0;
  unsigned combo = 0;  // the k-bit value whose combination the current row holds
  for (unsigned i = 1; i < (1u << k); ++i) {
    const int flip = __builtin_ctz(i);  // the bit in which Gray-code words i - 1 and i differ
    combo ^= 1u << flip;
    const word *src = M->rows[r + flip], *prev = T->rows[i - 1];
    word *dst = T->rows[i];
    for (wi_t j = w0; j + 1 < w; ++j) dst[j] = prev[j] ^ src[j];
    if (w > w0) dst[w - 1] = prev[w - 1] ^ (src[w - 1] & M->high_bitmask);
    L[combo] = (rci_t)i;
  }
}

extern "C" int m4ri_opt_k(int a, int b, int c) {
  // graycode.rs:44-56: "0.75 log_2(n) where n is min(a,b) for inversion and b for multiplication" (c != 0 = multiplication).
  // Only a hint on this side of the ABI: the tile kernels' tables are fixed at 8 bits by the LDS bank-row geometry.
  const int n = c != 0 ? b : (a < b ? a : b);
  if (n < 2) return 1;
  const int bits = 32 - __builtin_clz((unsigned)n);  // 1 + floor(log2 n)
  const int k = bits * 3 / 4;
  return k < 1 ? 1 : (k > 16 ? 16 : k);
}

// ---- compact binary wire format (the reference only has one-way serde JSON,
// m4ri-rust/src/friendly/binary_matrix.rs:10-35) ----
// file = "GF2M" | u32 version (1) | i32 nrows | i32 ncols | nrows * ceil(ncols/64) little-endian 64-bit words, rows dense,
// excess bits zero.
extern "C" int gf2_mzd_save(const char *path, mzd_t const *M) {
  FILE *f = std::fopen(path, "wb");
  if (!f) return -1;
  const uint32_t ver = 1;
  int ok = std::fwrite("GF2M", 1, 4, f) == 4 && std::fwrite(&ver, 4, 1, f) == 1 && std::fwrite(&M->nrows, 4, 1, f) == 1 &&
           std::fwrite(&M->ncols, 4, 1, f) == 1;
  for (rci_t i = 0; ok && i < M->nrows; ++i) {
    if (M->width > 1) ok = std::fwrite(M->rows[i], sizeof(word), (size_t)M->width - 1, f) == (size_t)M->width - 1;
    if (ok && M->width) {
      const word last = M->rows[i][M->width - 1] & M->high_bitmask;
      ok = std::fwrite(&last, sizeof(word), 1, f) == 1;
    }
  }
  return (std::fclose(f) == 0 && ok) ? 0 : -1;
}

extern "C" mzd_t *gf2_mzd_load(const char *path) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return nullptr;
  char magic[4];
  uint32_t ver = 0;
  rci_t r = 0, c = 0;
  mzd_t *M = nullptr;
  if (std::fread(magic, 1, 4, f) == 4 && std::memcmp(magic, "GF2M", 4) == 0 && std::fread(&ver, 4, 1, f) == 1 && ver == 1 &&
      std::fread(&r, 4, 1, f) == 1 && std::fread(&c, 4, 1, f) == 1 && r >= 0 && c >= 0) {
    M = mzd_init(r, c);
    for (rci_t i = 0; M && i < r; ++i) {
      if (M->width && std::fread(M->rows[i], sizeof(word), (size_t)M->width, f) != (size_t)M->width) {
        mzd_free(M);
        M = nullptr;
      } else if (M->width) {
        M->rows[i][M->width - 1] &= M->high_bitmask;
      }
    }
  }
  std::fclose(f);
  return M;
}
