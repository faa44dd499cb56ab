// m4ri_friendly.hpp -- header-only C++ mirror of m4ri-rust's friendly layer over the C ABI of libm4ri_hip.
//
// The reference's host side is Rust (m4ri-rust/src/friendly/{binary_matrix,binary_vector}.rs); no Rust toolchain
// exists in this environment, so the compiled-language host side above the C ABI is written in C++ with the same
// names, argument meaning and failure behaviour (a Rust panic = std::runtime_error here).  Only what the multiply
// path needs is mirrored; every product goes through
// mzd_mul / mzd_mul_m4rm / mzd_mul_naive exactly as `mul_impl!` selects them (binary_matrix.rs:53-95).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "m4ri_hip.h"

namespace m4ri_friendly {

struct Panic : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// Cargo features m4rm_mul / naive_mul / strassen_mul (m4ri-rust/Cargo.toml:29-34); default = Strassen
enum class MulStrategy { Strassen, M4rm, Naive };
inline MulStrategy &mul_strategy() {
  static MulStrategy s = MulStrategy::Strassen;
  return s;
}

class BinMatrix;

// binary_vector.rs:13-17: bit vector with LSB-first 64-bit blocks (vob::Vob storage convention)
class BinVector {
 public:
  BinVector() = default;
  static BinVector from_elem(size_t len, bool elem) {  // binary_vector.rs:55-58
    BinVector v;
    v.len_ = len;
    v.w_.assign((len + 63) / 64, elem ? ~0ull : 0ull);
    v.mask_last_block();
    return v;
  }
  static BinVector from_bools(const std::vector<bool> &b) {  // binary_vector.rs:61-65
    BinVector v = from_elem(b.size(), false);
    for (size_t i = 0; i < b.size(); ++i)
      if (b[i]) v.w_[i / 64] |= 1ull << (i % 64);
    return v;
  }
  static BinVector from_bytes(const std::vector<uint8_t> &bytes) {  // MSB-first per byte, binary_vector.rs:230-237
    BinVector v = from_elem(bytes.size() * 8, false);
    for (size_t i = 0; i < bytes.size() * 8; ++i)
      if ((bytes[i / 8] >> (7 - i % 8)) & 1) v.w_[i / 64] |= 1ull << (i % 64);
    return v;
  }
  static BinVector from_words(std::vector<uint64_t> w, size_t len) {
    BinVector v;
    v.w_ = std::move(w);
    v.len_ = len;
    v.w_.resize((len + 63) / 64);
    v.mask_last_block();
    return v;
  }
  size_t len() const { return len_; }
  bool get(size_t i) const { return (w_[i / 64] >> (i % 64)) & 1; }
  const std::vector<uint64_t> &get_storage() const { return w_; }
  uint32_t count_ones() const {  // binary_vector.rs:112-115
    uint32_t c = 0;
    for (uint64_t x : w_) c += (uint32_t)__builtin_popcountll(x);
    return c;
  }
  bool operator==(const BinVector &o) const { return len_ == o.len_ && w_ == o.w_; }
  BinVector operator+(const BinVector &o) const {  // binary_vector.rs:156-192
    if (len_ != o.len_) throw Panic("unequal length vectors");
    BinVector r = *this;
    for (size_t i = 0; i < w_.size(); ++i) r.w_[i] ^= o.w_[i];
    return r;
  }
  bool operator*(const BinVector &o) const {  // inner product, binary_vector.rs:194-215
    uint64_t acc = 0;
    const size_t k = w_.size() < o.w_.size() ? w_.size() : o.w_.size();
    for (size_t i = 0; i < k; ++i) acc ^= w_[i] & o.w_[i];
    return __builtin_parityll(acc);
  }
  inline BinMatrix as_matrix() const;         // binary_vector.rs:130-132
  inline BinMatrix as_column_matrix() const;  // binary_vector.rs:135-137

 private:
  void mask_last_block() {
    if (len_ % 64 && !w_.empty()) w_.back() &= (1ull << (len_ % 64)) - 1;
  }
  std::vector<uint64_t> w_;
  size_t len_ = 0;
};

// binary_matrix.rs:33-45: owns an mzd_t*, mzd_free on drop
class BinMatrix {
 public:
  explicit BinMatrix(mzd_t *m) : mzd_(m) {  // from_mzd, binary_matrix.rs:202-205
    if (!m) throw Panic("Can't be NULL");
  }
  BinMatrix(const BinMatrix &o) : mzd_(mzd_copy(nullptr, o.mzd_)) {}  // Clone, binary_matrix.rs:452-457
  BinMatrix(BinMatrix &&o) noexcept : mzd_(o.mzd_) { o.mzd_ = nullptr; }
  BinMatrix &operator=(BinMatrix o) {
    std::swap(mzd_, o.mzd_);
    return *this;
  }
  ~BinMatrix() {
    if (mzd_) mzd_free(mzd_);
  }

  static BinMatrix zero(size_t rows, size_t cols) {  // binary_matrix.rs:99-105
    if (rows == 0 || cols == 0) throw Panic("Can't create a 0 matrix");
    return BinMatrix(mzd_init((rci_t)rows, (rci_t)cols));
  }
  static BinMatrix from_slices(const std::vector<std::vector<uint64_t>> &rows, size_t rowlen) {  // :124-167
    if (rows.empty() || rowlen == 0) throw Panic("Can't create a 0 matrix");
    mzd_t *m = mzd_init((rci_t)rows.size(), (rci_t)rowlen);
    const size_t blocks = (rowlen + 63) / 64;
    for (size_t i = 0; i < rows.size(); ++i) {
      if (rows[i].size() * 64 < rowlen) {
        mzd_free(m);
        throw Panic("expected len " + std::to_string(rowlen) + " bits but got only " + std::to_string(rows[i].size()) + " blocks");
      }
      for (size_t b = 0; b < blocks; ++b) {
        uint64_t v = rows[i][b];
        if (b == rowlen / 64) v &= (1ull << (rowlen % 64)) - 1;  // tail masked, binary_matrix.rs:151-155
        m->rows[i][b] = v;
      }
    }
    return BinMatrix(m);
  }
  static BinMatrix new_(const std::vector<BinVector> &rows) {  // binary_matrix.rs:108-121
    std::vector<std::vector<uint64_t>> st;
    for (const auto &r : rows) st.push_back(r.get_storage());
    return from_slices(st, rows.at(0).len());
  }
  static BinMatrix random(size_t rows, size_t cols) {  // binary_matrix.rs:192-199
    mzd_t *m = mzd_init((rci_t)rows, (rci_t)cols);
    mzd_randomize(m);
    return BinMatrix(m);
  }
  static BinMatrix identity(size_t rows) {  // binary_matrix.rs:209-216
    mzd_t *m = mzd_init((rci_t)rows, (rci_t)rows);
    mzd_set_ui(m, 1);
    return BinMatrix(m);
  }

  size_t nrows() const { return (size_t)mzd_->nrows; }
  size_t ncols() const { return (size_t)mzd_->ncols; }
  bool bit(size_t r, size_t c) const { return (mzd_->rows[r][c / 64] >> (c % 64)) & 1; }  // binary_matrix.rs:364-368
  mzd_t *raw() const { return mzd_; }
  BinMatrix transposed() const { return BinMatrix(mzd_transpose(nullptr, mzd_)); }  // binary_matrix.rs:272-279
  BinMatrix augmented(const BinMatrix &o) const { return BinMatrix(mzd_concat(nullptr, mzd_, o.mzd_)); }
  BinMatrix stacked(const BinMatrix &o) const { return BinMatrix(mzd_stack(nullptr, mzd_, o.mzd_)); }
  size_t echelonize() { return (size_t)mzd_echelonize(mzd_, 0); }  // binary_matrix.rs:254-261
  size_t rank() const { return BinMatrix(*this).echelonize(); }    // binary_matrix.rs:246-252
  BinMatrix inverted() const { return BinMatrix(mzd_inv_m4ri(nullptr, mzd_, 0)); }  // :263-268, NULL -> "Can't be NULL"
  uint32_t count_ones() const {  // binary_matrix.rs:172-189
    if (!(nrows() == 1 || ncols() == 1)) throw Panic("only works on single row or single column matrices");
    uint32_t c = 0;
    for (size_t r = 0; r < nrows(); ++r)
      for (wi_t j = 0; j < mzd_->width; ++j) {
        uint64_t v = mzd_->rows[r][j];
        if (j == mzd_->width - 1) v &= mzd_->high_bitmask;
        c += (uint32_t)__builtin_popcountll(v);
      }
    return c;
  }
  BinVector as_vector() const {  // binary_matrix.rs:332-361
    if (nrows() != 1) {
      if (ncols() != 1) throw Panic("needs to have only one column or row");
      return transposed().as_vector();
    }
    std::vector<uint64_t> w(mzd_->rows[0], mzd_->rows[0] + mzd_->width);
    return BinVector::from_words(std::move(w), ncols());
  }
  // A * v^T with v as u64 words: always mzd_mul_naive (binary_matrix.rs:416-431)
  BinMatrix mul_slice(const std::vector<uint64_t> &other) const {
    if (!(ncols() <= other.size() * 64)) throw Panic("Mismatched sizes (too big)");
    BinMatrix vt = from_slices({other}, ncols()).transposed();
    return BinMatrix(mzd_mul_naive(nullptr, mzd_, vt.mzd_));
  }

  // operand cache (m4ri_hip.h): products whose operand is this matrix skip its upload until uncache() / destruction
  void cache_on_device() const {
    if (gf2_mzd_cache_on_device(mzd_) != 0) throw Panic(std::string("gf2_mzd_cache_on_device: ") + gf2_last_error());
  }
  void uncache() const { gf2_mzd_uncache(mzd_); }

  // serde wire format (feature "serde", binary_matrix.rs:10-35): exactly what serde_json::to_string prints
  // (test_serialize, binary_matrix.rs:693-699): {"matrix":{"rows":[{"len":N,"vec":[u64 words]},...]}}
  std::string to_json() const {
    std::string out = "{\"matrix\":{\"rows\":[";
    for (size_t r = 0; r < nrows(); ++r) {
      if (r) out += ',';
      out += "{\"len\":" + std::to_string(ncols()) + ",\"vec\":[";
      for (wi_t j = 0; j < mzd_->width; ++j) {
        uint64_t v = mzd_->rows[r][j];
        if (j == mzd_->width - 1) v &= mzd_->high_bitmask;
        if (j) out += ',';
        out += std::to_string(v);
      }
      out += "]}";
    }
    return out + "]}}";
  }

  bool operator==(const BinMatrix &o) const { return mzd_equal(mzd_, o.mzd_) == 1; }  // binary_matrix.rs:434-438
  bool operator!=(const BinMatrix &o) const { return !(*this == o); }
  BinMatrix operator+(const BinMatrix &o) const { return BinMatrix(mzd_add(nullptr, mzd_, o.mzd_)); }
  BinMatrix &operator+=(const BinMatrix &o) {
    mzd_add(mzd_, mzd_, o.mzd_);
    return *this;
  }
  // mul_impl! + impl Mul<&BinMatrix> for &BinMatrix (binary_matrix.rs:53-95, 459-472)
  BinMatrix operator*(const BinMatrix &o) const {
    mzd_t *p = nullptr;
    switch (mul_strategy()) {
      case MulStrategy::M4rm: p = mzd_mul_m4rm(nullptr, mzd_, o.mzd_, 0); break;
      case MulStrategy::Naive: p = mzd_mul_naive(nullptr, mzd_, o.mzd_); break;
      default: p = mzd_mul(nullptr, mzd_, o.mzd_, 0); break;
    }
    if (!p) throw Panic("Multiplication failed");
    return BinMatrix(p);
  }
  // impl Mul<&BinVector> for &BinMatrix: A * v^T (binary_matrix.rs:528-542)
  BinVector operator*(const BinVector &v) const { return mul_slice(v.get_storage()).as_vector(); }

 private:
  mzd_t *mzd_;
};

inline BinMatrix BinVector::as_matrix() const { return BinMatrix::new_({*this}); }
inline BinMatrix BinVector::as_column_matrix() const { return as_matrix().transposed(); }
// Solve A X = B, B modified in place; true if it succeeded (binary_matrix.rs:575-586; `a` is consumed there)
inline bool solve_left(BinMatrix a, BinMatrix &b) { return mzd_solve_left(a.raw(), b.raw(), 0, 1) == 0; }
// impl Mul<&BinMatrix> for &BinVector: v^T * A (binary_matrix.rs:552-563)
inline BinVector operator*(const BinVector &v, const BinMatrix &a) { return (v.as_matrix() * a).as_vector(); }

}  // namespace m4ri_friendly
