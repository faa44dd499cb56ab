// C++ mirror of the reference's own unit tests for the multiply path, over include/m4ri_friendly.hpp.
//   part "host": binary_matrix.rs:595-659,702-773 and binary_vector.rs:222-287 (no product, runs without a GPU)
//   part "mul" : binary_matrix.rs:662-686 (`mul`, `vecmul`) plus (A*B)*x == A*(B*x) under every mul strategy
// usage: test_friendly host|mul     exit code 0 = all assertions held
#include <cstdio>
#include <cstring>

#include "m4ri_friendly.hpp"
using namespace m4ri_friendly;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static int host_tests() {
  {  // new / identity (binary_matrix.rs:595-659)
    std::vector<BinVector> rows;
    for (int i = 0; i < 10; ++i) {
      std::vector<bool> b(10, false);
      b[i] = true;
      rows.push_back(BinVector::from_bools(b));
    }
    BinMatrix id = BinMatrix::new_(rows), gen = BinMatrix::identity(10);
    CHECK(id.nrows() == gen.nrows() && id.ncols() == gen.ncols());
    for (int i = 0; i < 10; ++i)
      for (int j = 0; j < 10; ++j) CHECK(id.bit(i, j) == gen.bit(i, j) && gen.bit(i, j) == (i == j));
    CHECK(id == gen);
  }
  {  // zero, random unequal (binary_matrix.rs:722-729, 758-762)
    BinMatrix z = BinMatrix::zero(10, 3);
    for (int i = 0; i < 10; ++i)
      for (int j = 0; j < 3; ++j) CHECK(!z.bit(i, j));
    CHECK(BinMatrix::random(100, 100) != BinMatrix::random(100, 100));
    bool threw = false;
    try { BinMatrix::zero(0, 3); } catch (const Panic &) { threw = true; }
    CHECK(threw);
  }
  for (int i = 1; i < 25; ++i) {  // as_vector round trips (binary_matrix.rs:702-719)
    BinMatrix c = BinMatrix::random(i, 1);
    BinVector v = c.as_vector();
    CHECK(v.len() == (size_t)i && c == v.as_column_matrix());
    BinMatrix r = BinMatrix::random(1, i);
    BinVector w = r.as_vector();
    CHECK(w.len() == (size_t)i && r == w.as_matrix());
  }
  {  // test_serialize (binary_matrix.rs:693-699)
    CHECK(BinMatrix::identity(3).to_json() ==
          "{\"matrix\":{\"rows\":[{\"len\":3,\"vec\":[1]},{\"len\":3,\"vec\":[2]},{\"len\":3,\"vec\":[4]}]}}");
  }
  {  // BinVector (binary_vector.rs:222-287)
    CHECK(BinVector::from_bytes({0xFF}).len() == 8);
    BinVector b = BinVector::from_bytes({0x80});
    CHECK(b.get(0) && !b.get(1));
    BinVector a = BinVector::from_elem(10, true), z = BinVector::from_elem(10, false);
    CHECK((z + z) == z && (a * z) == false && a.count_ones() == 10 && z.count_ones() == 0);
    CHECK(BinVector::from_bytes({0xA8}).count_ones() == 3);
    CHECK(a.count_ones() == a.as_matrix().count_ones() && a.count_ones() == a.as_column_matrix().count_ones());
  }
  return 0;
}

static int mul_tests() {
  const MulStrategy all[] = {MulStrategy::Strassen, MulStrategy::M4rm, MulStrategy::Naive};
  for (MulStrategy s : all) {
    mul_strategy() = s;
    {  // `mul` (binary_matrix.rs:662-670)
      BinMatrix prod = BinMatrix::identity(8) * BinMatrix::identity(8);
      CHECK(prod == BinMatrix::identity(8));
    }
    {  // `vecmul` (binary_matrix.rs:673-686)
      BinMatrix m1 = BinMatrix::identity(10);
      BinVector ones = BinVector::from_elem(10, true);
      CHECK((m1 * ones) == ones);
      CHECK((ones * m1) == ones);
      BinMatrix r = BinMatrix::random(10, 3);
      CHECK((ones * r).len() == 3);
    }
    {  // associativity with a vector and identity on ragged sizes
      BinMatrix A = BinMatrix::random(130, 257), B = BinMatrix::random(257, 300);
      BinVector x = BinMatrix::random(1, 300).as_vector();
      CHECK(((A * B) * x) == (A * (B * x)));
      CHECK((A * BinMatrix::identity(257)) == A);
      BinMatrix Bt = B.transposed(), At = A.transposed();
      CHECK((A * B).transposed() == (Bt * At));
    }
  }
  mul_strategy() = MulStrategy::Strassen;
  {  // elimination (binary_matrix.rs:246-268, 575-586): unit-triangular products are invertible
    const size_t n = 300;
    BinMatrix L = BinMatrix::identity(n), Um = BinMatrix::identity(n), R = BinMatrix::random(n, n);
    for (size_t i = 0; i < n; ++i)
      for (size_t j = 0; j < n; ++j)
        if (i != j && R.bit(i, j)) {
          mzd_t *t = (i > j ? L : Um).raw();
          t->rows[i][j / 64] |= 1ull << (j % 64);
        }
    BinMatrix A = L * Um;
    CHECK(A.rank() == n);
    BinMatrix Ai = A.inverted();
    CHECK((A * Ai) == BinMatrix::identity(n) && (Ai * A) == BinMatrix::identity(n));
    BinMatrix X0 = BinMatrix::random(n, 40), B = A * X0;
    CHECK(solve_left(A, B) && B == X0);
    BinMatrix Z = BinMatrix::zero(n, n);
    CHECK(Z.rank() == 0);
    BinMatrix thin = BinMatrix::random(200, 17) * BinMatrix::random(17, 200);
    CHECK(thin.rank() <= 17);
    bool threw = false;
    try {
      (void)thin.inverted();
    } catch (const Panic &) {
      threw = true;
    }
    CHECK(threw);
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const int rc = std::strcmp(argv[1], "mul") == 0 ? mul_tests() : host_tests();
  if (rc == 0) std::printf("%s ok\n", argv[1]);
  return rc;
}
