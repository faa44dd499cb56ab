"""CPU tests of the oracle itself: pinned against the reference's own known answers (identity and
vector products, bit order), the committed golden vectors (independent numpy product) and
cross-agreement of its four algorithms.  No GPU, no product code."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import gf2util as g

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALGOS = {
    "naive": g.o_mul_naive,
    "m4rm": g.o_mul_m4rm,
    "strassen": lambda a, b, m, l, n: g.o_mul_strassen(a, b, m, l, n, cutoff=64),
    "fast": g.o_mul_fast,
}


def identity_words(n):
    return g.bits_to_words(np.eye(n, dtype=np.uint8))


@pytest.mark.parametrize("algo", sorted(ALGOS))
def test_reference_known_answers(algo):
    """binary_matrix.rs:662-670 (I8*I8 == I8) and :673-686 (I10 * ones, ones * I10, 1x10 * 10x3 has 3 columns)."""
    f = ALGOS[algo]
    i8 = identity_words(8)
    assert np.array_equal(f(i8, i8, 8, 8, 8), i8)
    i10 = identity_words(10)
    ones_col = g.bits_to_words(np.ones((10, 1), dtype=np.uint8))
    ones_row = g.bits_to_words(np.ones((1, 10), dtype=np.uint8))
    assert np.array_equal(f(i10, ones_col, 10, 10, 1), ones_col)
    assert np.array_equal(f(ones_row, i10, 1, 10, 10), ones_row)
    r = g.random_words(10, 3, 5)
    assert f(ones_row, r, 1, 10, 3).shape == (1, 1)


def test_bit_order_matches_reference_layout():
    """mzd.rs:246-269: bit j of a row is bit j%64 of word j//64; serde test binary_matrix.rs:695-699:
    identity(3) rows are the words 1, 2, 4."""
    assert identity_words(3)[:, 0].tolist() == [1, 2, 4]
    b = np.zeros((1, 130), dtype=np.uint8)
    b[0, 65] = 1
    assert g.bits_to_words(b)[0].tolist() == [0, 2, 0]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))), ids=os.path.basename)
def test_golden_full(path):
    z = np.load(path)
    m, l, n = (int(x) for x in z["dims"])
    assert np.array_equal(g.random_words(m, l, 1), z["a"]) and np.array_equal(g.random_words(l, n, 2), z["b"])
    for name, f in ALGOS.items():
        assert np.array_equal(f(z["a"], z["b"], m, l, n), z["c"]), name
    if m * l * n <= 2_000_000:
        assert np.array_equal(g.o_mul_bits(z["a"], z["b"], m, l, n), z["c"])


def test_golden_digests():
    with open(os.path.join(GOLDEN, "digests.json")) as f:
        dig = json.load(f)
    for name in ("sq_1024", "sq_2048", "rect_3000x2100x2500"):
        d = dig[name]
        a, b = g.random_words(d["m"], d["l"], d["seed_a"]), g.random_words(d["l"], d["n"], d["seed_b"])
        for algo in ("m4rm", "fast", "strassen"):
            c = ALGOS[algo](a, b, d["m"], d["l"], d["n"])
            assert hashlib.sha256(c.tobytes()).hexdigest() == d["sha256_c"], (name, algo)


def test_generator_matches_c():
    lib = g.oracle()
    for (r, c) in [(1, 1), (5, 130), (64, 64), (33, 1000)]:
        a = g.random_words(r, c, 42)
        b = np.zeros_like(a)
        lib.oracle_fill_random(g.ptr(b), r, c, b.shape[1], 42)
        assert np.array_equal(a, b)
        if c % 64:
            assert not (a[:, -1] >> np.uint64(c % 64)).any()


@pytest.mark.parametrize("shape", [(1, 1, 1), (63, 64, 65), (127, 1, 129), (200, 333, 77), (640, 512, 384)])
def test_algebraic_identities(shape):
    m, l, n = shape
    a, b, c = g.random_words(m, l, 1), g.random_words(l, n, 2), g.random_words(l, n, 3)
    ab = g.o_mul_m4rm(a, b, m, l, n)
    # A(B+C) = AB + AC
    assert np.array_equal(g.o_mul_m4rm(a, b ^ c, m, l, n), ab ^ g.o_mul_m4rm(a, c, m, l, n))
    # (AB)^T = B^T A^T
    abt = g.o_transpose(ab, m, n)
    assert np.array_equal(abt, g.o_mul_m4rm(g.o_transpose(b, l, n), g.o_transpose(a, m, l), n, l, m))
    # excess bits of the last word stay zero
    if n % 64:
        assert not (ab[:, -1] >> np.uint64(n % 64)).any()
    # every k gives the same product
    for k in (1, 3, 8, 11):
        assert np.array_equal(g.o_mul_m4rm(a, b, m, l, n, k=k), ab)


def test_transpose_roundtrip_and_va():
    lib = g.oracle()
    for (r, c) in [(1, 1), (64, 64), (65, 63), (100, 257)]:
        w = g.random_words(r, c, 9)
        t = g.o_transpose(w, r, c)
        assert np.array_equal(g.words_to_bits(t, r), g.words_to_bits(w, c).T)
        assert np.array_equal(g.o_transpose(t, c, r), w)
    l, n = 300, 200
    v, a = g.random_words(1, l, 1), g.random_words(l, n, 2)
    out = np.zeros((1, g.width(n)), dtype=np.uint64)
    lib.oracle_mul_va(g.ptr(out), g.ptr(v), g.ptr(a), a.shape[1], l, n, 1)
    assert np.array_equal(out, g.o_mul_m4rm(v, a, 1, l, n))


def test_opt_k_follows_documented_rule():
    """graycode.rs:44-56: k ~ 0.75 * log2(n)."""
    lib = g.oracle()
    assert lib.oracle_opt_k(1000, 1024, 1000) == int(0.75 * 11)
    assert 1 <= lib.oracle_opt_k(1, 1, 1) <= 16


# ---- elimination oracle -----------------------------------------

def _py_rref(bits):
    """Independent bit-level Gauss-Jordan on a list-of-lists (small cases only)."""
    a = [list(map(int, r)) for r in bits]
    m, n = len(a), len(a[0]) if a else 0
    r, piv = 0, []
    for c in range(n):
        p = next((i for i in range(r, m) if a[i][c]), None)
        if p is None:
            continue
        a[r], a[p] = a[p], a[r]
        for i in range(m):
            if i != r and a[i][c]:
                a[i] = [x ^ y for x, y in zip(a[i], a[r])]
        piv.append(c)
        r += 1
        if r == m:
            break
    return np.array(a, dtype=np.uint8).reshape(m, n), piv


@pytest.mark.parametrize("m,n,seed", [(1, 1, 1), (5, 9, 2), (64, 64, 3), (70, 130, 4), (130, 70, 5), (100, 100, 6)])
def test_oracle_rref_matches_bit_level(m, n, seed):
    a = g.random_words(m, n, seed)
    red, rank, piv = g.o_echelonize(a, m, n, full=True)
    ref, rpiv = _py_rref(g.words_to_bits(a, n))
    assert rank == len(rpiv) and piv == rpiv
    assert np.array_equal(g.words_to_bits(red, n), ref)


def test_oracle_rank_deficient_and_upper_form():
    # rank <= 20 by construction: product of 90x20 and 20x150
    x, y = g.random_words(90, 20, 7), g.random_words(20, 150, 8)
    a = g.o_mul_naive(x, y, 90, 20, 150)
    red, rank, piv = g.o_echelonize(a, 90, 150, full=True)
    assert rank <= 20 and not red[rank:].any()
    up, rank2, piv2 = g.o_echelonize(a, 90, 150, full=False)
    assert rank2 == rank and piv2 == piv
    bits = g.words_to_bits(up, 150)
    for r, c in enumerate(piv2):  # echelon shape: first 1 of row r is at its pivot column, zeros below it
        assert bits[r, c] == 1 and not bits[r, :c].any() and not bits[r + 1:, c].any()
    # same row space: the reduced form of the upper form is the reduced form
    assert np.array_equal(g.o_echelonize(up, 90, 150, full=True)[0], red)


def test_oracle_inverse_and_solve():
    n = 150
    a = g.random_words(n, n, 11)
    inv = g.o_inverse(a, n)
    while inv is None:  # random matrices are invertible with probability ~0.29
        a[0, 0] ^= np.uint64(1)
        a = g.random_words(n, n, int(a[0, 0]) % 1000 + 12)
        inv = g.o_inverse(a, n)
    ident = g.o_mul_naive(a, inv, n, n, n)
    assert np.array_equal(g.words_to_bits(ident, n), np.eye(n, dtype=np.uint8))
    assert g.o_inverse(np.zeros_like(a), n) is None
    # A X = B with a known solution
    m, n, k = 120, 80, 33
    a = g.random_words(m, n, 13)
    x0 = g.random_words(n, k, 14)
    b = np.zeros((m, g.width(k)), dtype=np.uint64)
    b[:m] = g.o_mul_naive(a, x0, m, n, k)
    x, ok = g.o_solve_left(a, m, n, b, m, k)
    assert ok and np.array_equal(g.o_mul_naive(a, x[:n], m, n, k), b) and not x[n:].any()
    b[m - 1, 0] ^= np.uint64(1)  # rank(A) = 80 < 120: a perturbed right-hand side is (almost surely) inconsistent
    assert g.o_solve_left(a, m, n, b, m, k)[1] is False


# ---- committed elimination fixtures (tests/golden/elim, independent numpy elimination) ----------

ELIM = sorted(glob.glob(os.path.join(GOLDEN, "elim", "*.npz")))


@pytest.mark.parametrize("path", ELIM, ids=[os.path.basename(p)[:-4] for p in ELIM])
def test_oracle_matches_elimination_fixtures(path):
    d = np.load(path)
    name = os.path.basename(path)
    if name.startswith("rref"):
        m, n = (int(x) for x in d["shape"])
        red, rank, piv = g.o_echelonize(d["a"], m, n, full=True)
        assert rank == len(d["pivots"]) and piv == list(d["pivots"]) and np.array_equal(red, d["rref"])
    elif name.startswith("inverse"):
        n = int(d["shape"][0])
        assert np.array_equal(g.o_inverse(d["a"], n), d["inv"])
    else:
        m, n, k = (int(x) for x in d["shape"])
        x, ok = g.o_solve_left(d["a"], m, n, d["b"], m, k)
        assert ok and np.array_equal(x, d["x"])
        assert g.o_solve_left(d["a"], m, n, d["b_inconsistent"], m, k)[1] == (not bool(d["inconsistent"][0]))


# ---- full-size digests of the BASELINE configurations (tests/golden/make_golden_large.py) -------------------

def _large():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "digests_large.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("v", [1, 64, 128, 256])
def test_oracle_reproduces_full_size_lpn_digest(v):
    """BASELINE config 5 at 2^20 rows: the oracle's Four-Russians product hashes to the digest of the independent numpy product."""
    import hashlib
    d = _large()["lpn_1048576x256x%d" % v]
    m, l, n = d["m"], d["l"], d["n"]
    a, b = g.random_words(m, l, d["seed_a"]), g.random_words(l, n, d["seed_b"])
    assert hashlib.sha256(g.o_mul_m4rm(a, b, m, l, n).tobytes()).hexdigest() == d["sha256_c"]
    if v == 1:
        assert hashlib.sha256(g.o_mul_naive(a, b, m, l, n).tobytes()).hexdigest() == d["sha256_c"]


def test_full_size_square_digest_rows_against_plain_m4rm():
    """sq_32768: the committed sample rows of the oracle_mul_fast product equal plain Four Russians on those rows."""
    import hashlib
    d = _large()["sq_32768"]
    n, rows = d["n"], d["sample_rows"]
    a = g.random_words(n, n, d["seed_a"])[rows]
    b = g.random_words(n, n, d["seed_b"])
    got = g.o_mul_m4rm(np.ascontiguousarray(a), b, len(rows), n, n)
    assert hashlib.sha256(got.tobytes()).hexdigest() == d["sample_rows_sha256"]
    assert set(_large()) >= {"lpn_1048576x256x1", "lpn_1048576x256x64", "lpn_1048576x256x256", "sq_32768", "sq_65536"}
