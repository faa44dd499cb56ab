// bench_shapes -- the reference's own benchmark shapes (m4ri-rust/benches/binary_matrix.rs:11-76) timed through the C ABI of
// libm4ri_hip.so, beside the CPU port (oracle/libgf2oracle.so) doing the same sequence of library calls on one core.
// Development / measurement tool (SURVEY.md section 8c "harness rows"): the Rust benches cannot run here (no rustc), so each
// #[bench] is restated as the C calls its body makes through m4ri-sys:
//   &m1 * &m2            mzd_mul(NULL, a, b, 0) [default mul_impl!, binary_matrix.rs:63-74,459-472]  + Drop -> mzd_free
//   &v * &m              as_matrix = mzd_init(1, len) + row fill; mzd_mul(NULL, v, m, 0); as_vector = word copy; 2 x mzd_free
//   &m * &v / mul_slice  from_slices (mzd_init + row fill), mzd_transpose(NULL, .), mzd_mul_naive(NULL, m, vT),
//                        as_vector: mzd_transpose(NULL, .) + word copy; 4 x mzd_free            [binary_matrix.rs:416-431,332-361]
//   as_vector_*          the conversions alone
// Output: one JSON object per line: {"bench":..., "us_per_iter_dropin":..., "us_per_iter_forced_device":..., "us_per_iter_cpu_port":...}
//   dropin = default size dispatch (M4RI_HIP_HOST_SMALL_WORK unset), forced_device = M4RI_HIP_HOST_SMALL_WORK=0.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../include/m4ri_hip.h"
extern "C" {
#include "../oracle/gf2_oracle.h"
}

static double time_us(const std::function<void()> &f, int min_iters, double min_seconds) {
  f();
  f();
  int iters = 0;
  const auto t0 = std::chrono::steady_clock::now();
  double el = 0;
  do {
    for (int i = 0; i < min_iters; ++i) f();
    iters += min_iters;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  } while (el < min_seconds);
  return el / iters * 1e6;
}

static mzd_t *rnd(int r, int c, unsigned seed) {
  mzd_t *m = mzd_init(r, c);
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < m->width; ++j) {
      word v = oracle_splitmix64(seed, (uint64_t)i * m->width + j);
      if (j == m->width - 1) v &= m->high_bitmask;
      m->rows[i][j] = v;
    }
  return m;
}
static std::vector<uint64_t> dense(const mzd_t *m) {
  std::vector<uint64_t> d((size_t)m->nrows * m->width);
  for (int i = 0; i < m->nrows; ++i) memcpy(&d[(size_t)i * m->width], m->rows[i], (size_t)m->width * 8);
  return d;
}

struct Bench {
  std::string name;
  std::function<void()> lib, cpu;
};

static volatile uint64_t sink;

int main() {
  std::vector<Bench> benches;
  auto add_mul = [&](const char *name, int a, int b, int d) {
    mzd_t *m1 = rnd(a, b, 1), *m2 = rnd(b, d, 2);
    auto A = new std::vector<uint64_t>(dense(m1));
    auto B = new std::vector<uint64_t>(dense(m2));
    const int wa = m1->width, wb = m2->width;
    benches.push_back({name,
                       [=] {
                         mzd_t *c = mzd_mul(nullptr, m1, m2, 0);
                         if (!c) abort();
                         sink = c->rows[0][0];
                         mzd_free(c);
                       },
                       [=] {
                         uint64_t *c = (uint64_t *)calloc((size_t)a * wb, 8);  // mzd_init of the result
                         oracle_mul_strassen(c, wb, A->data(), wa, B->data(), wb, a, b, d, 0, 1);
                         sink = c[0];
                         free(c);
                       }});
  };
  {  // as_vector_column / as_vector_column_transpose / as_vector_row: 1000 x 1 and 1 x 1000 (binary_matrix.rs:11-27)
    mzd_t *col = rnd(1000, 1, 3), *row = rnd(1, 1000, 4);
    auto C = new std::vector<uint64_t>(dense(col));
    benches.push_back({"as_vector_column (= transposed + row copy, binary_matrix.rs:332-336)",
                       [=] {
                         mzd_t *t = mzd_transpose(nullptr, col);
                         uint64_t buf[16];
                         memcpy(buf, t->rows[0], 16 * 8);
                         sink = buf[0];
                         mzd_free(t);
                       },
                       [=] {
                         uint64_t *t = (uint64_t *)calloc(16, 8);
                         oracle_transpose(t, 16, C->data(), 1, 1000, 1);
                         sink = t[0];
                         free(t);
                       }});
    benches.push_back({"as_vector_row",
                       [=] {
                         uint64_t buf[16];
                         memcpy(buf, row->rows[0], 16 * 8);
                         sink = buf[0];
                       },
                       [=] {
                         uint64_t buf[16];
                         memcpy(buf, row->rows[0], 16 * 8);
                         sink = buf[0];
                       }});
  }
  {  // vector_matrix: 1000-bit v * (1000 x 64)   (binary_matrix.rs:30-34, 552-563)
    mzd_t *m = rnd(1000, 64, 5), *v = rnd(1, 1000, 6);
    auto M = new std::vector<uint64_t>(dense(m));
    auto V = new std::vector<uint64_t>(dense(v));
    benches.push_back({"vector_matrix 1x1000 * 1000x64",
                       [=] {
                         mzd_t *vm = mzd_init(1, 1000);
                         memcpy(vm->rows[0], v->rows[0], 16 * 8);
                         mzd_t *c = mzd_mul(nullptr, vm, m, 0);
                         sink = c->rows[0][0];
                         mzd_free(c);
                         mzd_free(vm);
                       },
                       [=] {
                         uint64_t *vm = (uint64_t *)calloc(16, 8), *c = (uint64_t *)calloc(1, 8);
                         memcpy(vm, V->data(), 16 * 8);
                         oracle_mul_strassen(c, 1, vm, 16, M->data(), 1, 1, 1000, 64, 0, 1);
                         sink = c[0];
                         free(c);
                         free(vm);
                       }});
  }
  auto add_matvec = [&](const char *name, int rows, int cols, int vbits) {  // mul_slice path (binary_matrix.rs:416-431)
    mzd_t *m = rnd(rows, cols, 7), *v = rnd(1, vbits, 8);
    auto M = new std::vector<uint64_t>(dense(m));
    auto V = new std::vector<uint64_t>(dense(v));
    const int wc = m->width;
    benches.push_back({name,
                       [=] {
                         mzd_t *vm = mzd_init(1, cols);
                         memcpy(vm->rows[0], v->rows[0], (size_t)wc * 8);
                         vm->rows[0][wc - 1] &= vm->high_bitmask;
                         mzd_t *vt = mzd_transpose(nullptr, vm);
                         mzd_t *c = mzd_mul_naive(nullptr, m, vt);
                         mzd_t *ct = mzd_transpose(nullptr, c);
                         sink = ct->rows[0][0];
                         mzd_free(ct);
                         mzd_free(c);
                         mzd_free(vt);
                         mzd_free(vm);
                       },
                       [=] {
                         uint64_t *vm = (uint64_t *)calloc(wc, 8), *vt = (uint64_t *)calloc(cols, 8), *c = (uint64_t *)calloc(rows, 8),
                                  *ct = (uint64_t *)calloc((rows + 63) / 64, 8);
                         memcpy(vm, V->data(), (size_t)wc * 8);
                         oracle_transpose(vt, 1, vm, wc, 1, cols);
                         oracle_mul_naive(c, 1, M->data(), wc, vt, 1, rows, cols, 1);
                         oracle_transpose(ct, (rows + 63) / 64, c, 1, rows, 1);
                         sink = ct[0];
                         free(ct);
                         free(c);
                         free(vt);
                         free(vm);
                       }});
  };
  add_matvec("matrix_vector 64x1000 * v", 64, 1000, 1000);
  add_matvec("matrix_mul_slice 34x128 * 128 bits", 34, 128, 500);
  add_mul("matrix_multiply_10x10_10x10", 10, 10, 10);
  add_mul("matrix_multiply_100x10_10x10", 100, 10, 10);
  add_mul("matrix_multiply_100x10_10x100", 100, 10, 100);
  add_mul("matrix_multiply_1000x64_64x1000", 1000, 64, 1000);
  add_mul("matrix_multiply_1000x10_10x1000", 1000, 10, 1000);
  add_mul("matrix_multiply_1000x64_64x1", 1000, 64, 1);
  add_mul("matrix_multiply_1x64_64x100 (1 x 64 * 64 x 1000)", 1, 64, 1000);
  add_mul("matrix_multiply_10x1000_1000x10", 10, 1000, 10);

  for (auto &b : benches) {
    unsetenv("M4RI_HIP_HOST_SMALL_WORK");
    const double lib = time_us(b.lib, 50, 0.2);
    setenv("M4RI_HIP_HOST_SMALL_WORK", "0", 1);
    const double dev = time_us(b.lib, 20, 0.2);
    unsetenv("M4RI_HIP_HOST_SMALL_WORK");
    const double cpu = time_us(b.cpu, 50, 0.2);
    printf("{\"bench\": \"%s\", \"us_per_iter_dropin\": %.2f, \"us_per_iter_forced_device\": %.2f, \"us_per_iter_cpu_port\": %.2f, "
           "\"dropin_over_cpu_port\": %.2f}\n",
           b.name.c_str(), lib, dev, cpu, lib / cpu);
    fflush(stdout);
  }
  return 0;
}
