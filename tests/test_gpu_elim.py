"""GPU parity tests for the elimination entry points: mzd_echelonize*, mzd_inv_m4ri,
mzd_solve_left and the device forms, through the C ABI, against the CPU oracle.  Reduced row echelon forms, inverses and
solutions with free variables 0 are unique, so every comparison is bit-exact; the non-reduced form (full = 0) is
checked through what is defined about it (rank, pivot columns, echelon shape, row space)."""
import os

import numpy as np
import pytest

import gf2util as g

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg(built):
    import m4ri_rust_amd as p
    from m4ri_rust_amd import device
    device.require_gpu()
    return p


@pytest.fixture(scope="module")
def dev(pkg):
    from m4ri_rust_amd import device
    return device


@pytest.fixture(params=[32, 2], ids=["block2048", "block128"])
def block_words(request):
    """Column-block width of the elimination; the small setting puts many blocks into small matrices."""
    old = os.environ.get("M4RI_HIP_ELIM_BLOCK_WORDS")
    os.environ["M4RI_HIP_ELIM_BLOCK_WORDS"] = str(request.param)
    yield request.param
    if old is None:
        del os.environ["M4RI_HIP_ELIM_BLOCK_WORDS"]
    else:
        os.environ["M4RI_HIP_ELIM_BLOCK_WORDS"] = old


def _low_rank(m, n, r, seed):
    return g.o_mul_naive(g.random_words(m, r, seed), g.random_words(r, n, seed + 1), m, r, n)


def _host_rref(pkg, a, n, full=True):
    M = pkg.BinMatrix.from_words(a, n)
    rank = M.echelonize(full=full)
    return M.to_words(), rank


SHAPES = [(1, 1), (1, 200), (200, 1), (64, 64), (65, 63), (100, 100), (70, 130), (130, 70), (513, 1030), (1030, 513),
          (1000, 1000), (2048, 2048), (300, 5000), (5000, 300), (3000, 2500)]


@pytest.mark.parametrize("m,n", SHAPES)
def test_rref_random(pkg, block_words, m, n):
    a = g.random_words(m, n, 100 + m + n)
    got, rank = _host_rref(pkg, a, n)
    ref, orank, _ = g.o_echelonize(a, m, n, full=True)
    assert rank == orank
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("m,n,r", [(300001, 200, 200), (270000, 130, 70), (33000, 193, 193)])
@pytest.mark.parametrize("full", [True, False])
def test_rref_tall(pkg, m, n, r, full):
    """More rows than one turn of the update workgroups covers (255 x 1024 rows in the look-ahead's one-row-per-lane pass of the next
    word column, 255 x 128 per pass of the waves): every workgroup walks its contiguous piece in several turns; the pivots of a
    low-rank tall matrix sit far apart, so the search scans past its first 256 rows too."""
    a = g.random_words(m, n, 31 + m) if r == n else _low_rank(m, n, r, m)
    a[: m // 2] = 0 if r != n else a[: m // 2]  # (low rank: the upper half is empty, every pivot comes from far down)
    got, rank = _host_rref(pkg, a, n, full=full)
    ref, orank, _ = g.o_echelonize(a, m, n, full=True)
    assert rank == orank <= r
    if full:
        assert np.array_equal(got, ref)
    else:  # upper form: same rank and, reduced on the host, the same reduced form
        again, rank2, _ = g.o_echelonize(got, m, n, full=True)
        assert rank2 == rank and np.array_equal(again, ref)


@pytest.mark.parametrize("m,n,r", [(200, 300, 17), (1500, 1200, 64), (1500, 1200, 65), (2500, 3000, 700), (4096, 4096, 1)])
def test_rref_rank_deficient(pkg, block_words, m, n, r):
    a = _low_rank(m, n, r, 7 * r + m)
    got, rank = _host_rref(pkg, a, n)
    ref, orank, _ = g.o_echelonize(a, m, n, full=True)
    assert rank == orank <= r
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("m,n,r", [(10, 100000, 10), (10, 100000, 4), (18, 66000, 18), (3, 70000, 2)])
@pytest.mark.parametrize("full", [True, False])
def test_rref_short_and_wide(pkg, m, n, r, full):
    """Rows wider than 1024 words with a handful of rows (the one-workgroup kernel swapped one word per
    thread only); rank-deficient cases included."""
    a = g.random_words(m, n, 900 + m + r) if r == m else _low_rank(m, n, r, 900 + m + r)
    got, rank = _host_rref(pkg, a, n, full=full)
    ref, orank, opiv = g.o_echelonize(a, m, n, full=True)
    assert rank == orank
    if full:
        assert np.array_equal(got, ref)
    else:  # same row space and pivots: reducing it fully gives the unique reduced form
        again, rank2, _ = g.o_echelonize(got, m, n, full=True)
        assert rank2 == rank and np.array_equal(again, ref)


@pytest.mark.parametrize("m,n,k", [(10, 10, 100000), (16, 12, 70000)])
def test_solve_left_wide_rhs_small_kernel(pkg, m, n, k):
    """[A | B] with a right-hand side wider than 1024 words and a column limit of n: the route on which the
    one-workgroup kernel sees rows of more than 1024 words."""
    a = g.random_words(m, n, 930 + m)
    a[0], a[m - 1] = a[m - 1].copy(), a[0].copy()
    x0 = g.random_words(n, k, 931)
    b = g.o_mul_naive(a, x0, m, n, k)
    A, B = pkg.BinMatrix.from_words(a, n), pkg.BinMatrix.from_words(b, k)
    ref, ok = g.o_solve_left(a, m, n, b, m, k)
    assert pkg.solve_left(A, B) is ok
    if ok:
        assert np.array_equal(B.to_words(), ref)


def test_rref_structured(pkg, block_words):
    n = 700
    ident = g.bits_to_words(np.eye(n, dtype=np.uint8))
    got, rank = _host_rref(pkg, ident, n)
    assert rank == n and np.array_equal(got, ident)
    zero = np.zeros_like(ident)
    got, rank = _host_rref(pkg, zero, n)
    assert rank == 0 and not got.any()
    # reversed identity: every pivot comes from the last active row
    rev = g.bits_to_words(np.eye(n, dtype=np.uint8)[::-1].copy())
    got, rank = _host_rref(pkg, rev, n)
    assert rank == n and np.array_equal(got, ident)
    # leading zero columns, duplicated rows, an all-ones block
    bits = g.words_to_bits(g.random_words(900, 1000, 5), 1000)
    bits[:, :130] = 0
    bits[450:] = bits[:450]
    bits[100:164, 500:564] = 1
    a = g.bits_to_words(bits)
    got, rank = _host_rref(pkg, a, 1000)
    ref, orank, _ = g.o_echelonize(a, 900, 1000, full=True)
    assert rank == orank and np.array_equal(got, ref)


def _structured_cases():
    """Inputs for the blocked path (more than 1024 rows) that steer the 64-column step into its rarer branches (round 5): a search
    that does not find its 64 pivots among the first 256 candidate rows (reads on behind the previous publication's stash, after
    waiting for the whole update; chosen rows beyond the first 256), steps with fewer than 64 pivots (the one-row-per-lookup
    update; the general insertion loop behind the in-lane one), leading runs of flagged rows, rows that cancel."""
    rng = np.random.default_rng(77)
    out = {}
    base = g.words_to_bits(g.random_words(500, 2000, 41), 2000)
    out["every-row-seven-times"] = np.repeat(base, 7, axis=0)[:3300]                      # 256 candidates hold ~37 independent rows
    z = np.zeros((3000, 1500), dtype=np.uint8)
    z[::3] = g.words_to_bits(g.random_words(1000, 1500, 42), 1500)
    out["two-zero-rows-between"] = z                                                      # sparse candidates, many passes
    d = g.words_to_bits(g.random_words(2600, 900, 43), 900)
    out["column-pairs-equal"] = np.repeat(d, 2, axis=1)                                   # at most 32 pivots per word column
    out["reversed-identity"] = np.eye(2100, dtype=np.uint8)[::-1].copy()                  # every pivot in the last active row
    lo = g.words_to_bits(_low_rank(2500, 1900, 90, 44), 1900)
    lo[rng.permutation(2500)[:700]] = 0
    out["low-rank-with-zero-rows"] = lo
    t = np.tril(np.ones((1800, 1800), dtype=np.uint8))
    out["lower-triangular-ones"] = t                                                      # row i = row i-1 + unit: long dependency chains
    w = g.words_to_bits(g.random_words(2048, 2048, 45), 2048)
    w[:, 64:128] = 0
    w[:, 700:1000] = 0
    w[1000:1400] = w[200:600] ^ w[600:1000]
    out["zero-column-bands-and-sums"] = w
    # tall enough (>= 192 rows per update workgroup) for the publication to go ahead of the update's end: priority rows, flags
    # copied into LDS up front, and -- with repeated rows -- the fall-back to the whole update when the search leaves its first pass
    tb = g.words_to_bits(g.random_words(9000, 190, 46), 190)
    out["tall-every-row-seven-times"] = np.repeat(tb, 7, axis=0)[:60000]
    tc = g.words_to_bits(g.random_words(52000, 128, 47), 128)
    tc[rng.random(52000) < 0.5] = 0
    out["tall-column-pairs-zero-rows"] = np.repeat(tc, 2, axis=1)
    return out


_STRUCT = None


@pytest.mark.parametrize("name", ["every-row-seven-times", "two-zero-rows-between", "column-pairs-equal", "reversed-identity",
                                  "low-rank-with-zero-rows", "lower-triangular-ones", "zero-column-bands-and-sums",
                                  "tall-every-row-seven-times", "tall-column-pairs-zero-rows"])
@pytest.mark.parametrize("full", [True, False], ids=["rref", "upper"])
def test_rref_structured_blocked(pkg, block_words, name, full):
    global _STRUCT
    if _STRUCT is None:
        _STRUCT = _structured_cases()
    bits = _STRUCT[name]
    m, n = bits.shape
    a = g.bits_to_words(bits)
    got, rank = _host_rref(pkg, a, n, full=full)
    ref, orank, _ = g.o_echelonize(a, m, n, full=True)
    assert rank == orank
    if full:
        assert np.array_equal(got, ref)
    else:  # same row space and pivots: reducing it fully gives the unique reduced form
        again, rank2, _ = g.o_echelonize(got, m, n, full=True)
        assert rank2 == rank and np.array_equal(again, ref)


@pytest.mark.parametrize("m,n,r", [(100, 100, 100), (1200, 1500, 1500), (1500, 1200, 300), (3000, 2500, 2500)])
def test_upper_echelon_form(pkg, block_words, m, n, r):
    """full = 0 (what BinMatrix::echelonize passes, binary_matrix.rs:259)."""
    a = _low_rank(m, n, r, 31) if r < min(m, n) else g.random_words(m, n, 32)
    got, rank = _host_rref(pkg, a, n, full=False)
    ref, orank, piv = g.o_echelonize(a, m, n, full=True)
    assert rank == orank
    bits = g.words_to_bits(got, n)
    assert not bits[rank:].any()
    first = np.argmax(bits[:rank], axis=1)  # leading 1 of each non-zero row
    assert list(first) == piv  # same pivot columns, strictly increasing
    for r_, c in enumerate(piv):
        assert not bits[r_ + 1:, c].any()
    assert np.array_equal(g.o_echelonize(got, m, n, full=True)[0], ref)  # same row space


def test_rank_and_variants(pkg):
    L = pkg._lib.lib()
    a = _low_rank(777, 900, 123, 3)
    A = pkg.BinMatrix.from_words(a, 900)
    before = A.to_words().copy()
    assert A.rank() == g.o_echelonize(a, 777, 900)[1]
    assert np.array_equal(A.to_words(), before)  # rank() works on a clone (binary_matrix.rs:250-252)
    ref, orank, _ = g.o_echelonize(a, 777, 900, full=True)
    for fn, extra in ((L.mzd_echelonize_m4ri, (1, 0)), (L.mzd_echelonize_pluq, (1,))):
        B = pkg.BinMatrix.from_words(a, 900)
        assert fn(B.mzd, *extra) == orank
        assert np.array_equal(B.to_words(), ref)


def test_echelonize_window(pkg):
    L = pkg._lib.lib()
    big = pkg.BinMatrix.from_words(g.random_words(400, 600, 41), 600)
    before = g.words_to_bits(big.to_words(), 600)
    W = L.mzd_init_window(big.mzd, 30, 128, 330, 128 + 200)  # 300 x 200, ragged tail inside the parent
    rank = L.mzd_echelonize(W, 1)
    after = g.words_to_bits(big.to_words(), 600)
    sub = g.bits_to_words(before[30:330, 128:328])
    ref, orank, _ = g.o_echelonize(sub, 300, 200, full=True)
    expect = before.copy()
    expect[30:330, 128:328] = g.words_to_bits(ref, 200)
    assert rank == orank and np.array_equal(after, expect)
    L.mzd_free(W)


def _invertible(n, seed):
    """Product of a unit lower and a unit upper triangular random matrix."""
    bits = g.words_to_bits(g.random_words(n, n, seed), n)
    lo = np.tril(bits, -1) | np.eye(n, dtype=np.uint8)
    up = np.triu(g.words_to_bits(g.random_words(n, n, seed + 1), n), 1) | np.eye(n, dtype=np.uint8)
    return g.o_mul_naive(g.bits_to_words(lo), g.bits_to_words(up), n, n, n)


@pytest.mark.parametrize("n", [1, 64, 100, 1000, 2049])
def test_inverse(pkg, block_words, n):
    a = _invertible(n, 50 + n)
    A = pkg.BinMatrix.from_words(a, n)
    inv = A.inverted()
    assert np.array_equal(inv.to_words(), g.o_inverse(a, n))
    assert (A * inv) == pkg.BinMatrix.identity(n)


def test_inverse_singular(pkg):
    a = _low_rank(300, 300, 299, 9)
    with pytest.raises(pkg.PanicError, match="Can't be NULL"):
        pkg.BinMatrix.from_words(a, 300).inverted()
    L = pkg._lib.lib()
    dst = pkg.BinMatrix.zero(300, 300)
    assert not L.mzd_inv_m4ri(dst.mzd, pkg.BinMatrix.from_words(a, 300).mzd, 0)


@pytest.mark.parametrize("m,n,k", [(100, 100, 1), (1200, 800, 70), (2048, 2048, 2048), (900, 900, 5000)])
def test_solve_left(pkg, block_words, m, n, k):
    a = g.random_words(m, n, 61)
    x0 = g.random_words(n, k, 62)
    b = g.o_mul_naive(a, x0, m, n, k)
    A, B = pkg.BinMatrix.from_words(a, n), pkg.BinMatrix.from_words(b, k)
    assert pkg.solve_left(A, B) is True
    ref, ok = g.o_solve_left(a, m, n, b, m, k)
    assert ok and np.array_equal(B.to_words(), ref)
    assert np.array_equal(g.o_mul_naive(a, B.to_words()[:n], m, n, k), b)
    assert np.array_equal(A.to_words(), g.o_echelonize(a, m, n, full=True)[0])  # "A ... (overwritten)"


def test_solve_left_underdetermined_rows_and_inconsistent(pkg):
    # rank-deficient A, more rows in B than in A: the extra rows are cleared, free variables are 0
    m, n, k, brows = 500, 400, 90, 640
    a = _low_rank(m, n, 150, 71)
    x0 = g.random_words(n, k, 72)
    b = np.zeros((brows, g.width(k)), dtype=np.uint64)
    b[:m] = g.o_mul_naive(a, x0, m, n, k)
    b[m:] = g.random_words(brows - m, k, 73)  # ignored by the solver
    A, B = pkg.BinMatrix.from_words(a, n), pkg.BinMatrix.from_words(b, k)
    assert pkg.solve_left(A, B) is True
    ref, ok = g.o_solve_left(a, m, n, b, brows, k)
    assert ok and np.array_equal(B.to_words(), ref)
    assert np.array_equal(g.o_mul_naive(a, B.to_words()[:n], m, n, k), b[:m])
    # inconsistent system
    b2 = b.copy()
    b2[:m] = g.random_words(m, k, 74)
    A, B = pkg.BinMatrix.from_words(a, n), pkg.BinMatrix.from_words(b2, k)
    assert pkg.solve_left(A, B) is False
    assert g.o_solve_left(a, m, n, b2, brows, k)[1] is False
    # without the check the call reports success (solve.rs:19-21: output then undefined)
    L = pkg._lib.lib()
    A, B = pkg.BinMatrix.from_words(a, n), pkg.BinMatrix.from_words(b2, k)
    assert L.mzd_solve_left(A.mzd, B.mzd, 0, 0) == 0


def test_device_echelonize_limit_and_pivots(pkg, dev, block_words):
    m, n, extra = 1500, 1000, 700
    a = _low_rank(m, n + extra, 600, 81)
    D = dev.DMat.from_words(a, n + extra)
    rank, piv = dev.echelonize(D, full=True, ncols_limit=n)
    ref, orank, opiv = g.o_echelonize(a, m, n + extra, full=True, limit=n)
    assert rank == orank and piv == opiv
    assert np.array_equal(D.to_words(), ref)


def test_device_inverse(pkg, dev):
    n = 3000
    a = _invertible(n, 91)
    A = dev.DMat.from_words(a, n)
    inv = dev.inverse(A)
    assert inv is not None and np.array_equal(inv.to_words(), g.o_inverse(a, n))
    assert dev.inverse(dev.DMat.from_words(_low_rank(500, 500, 20, 92), 500)) is None


@pytest.mark.parametrize("n", [8192, 16384])
def test_large_transformation_property(pkg, dev, n):
    """[A | I] -> [R | E] with E*A = R, R the reduced echelon form: checked with the device product (no oracle at
    this size), plus the shape of R."""
    A = dev.DMat.random(n, n, 7)
    T = dev.DMat(n, 2 * n)
    aw = A.to_words()
    t = np.zeros((n, 2 * n // 64), dtype=np.uint64)
    t[:, : n // 64] = aw
    t[:, n // 64:] = g.bits_to_words(np.eye(n, dtype=np.uint8))
    T = dev.DMat.from_words(t, 2 * n)
    rank, piv = dev.echelonize(T, full=True, ncols_limit=n)
    out = T.to_words()
    R, E = np.ascontiguousarray(out[:, : n // 64]), np.ascontiguousarray(out[:, n // 64:])
    prod = dev.mul(dev.DMat.from_words(E, n), A, algo="m4rm").to_words()
    assert np.array_equal(prod, R)
    assert n - 70 < rank <= n and len(piv) == rank and piv == sorted(piv)
    bits = g.words_to_bits(R, n)
    sub = bits[:rank][:, piv]
    assert np.array_equal(sub, np.eye(rank, dtype=np.uint8)) and not bits[rank:].any()


def test_full_size_transformation_property(pkg, dev):
    """65536 x 65536 (BASELINE's headline dimension), entirely on the device: [A | I] -> [R | E], then E*A == R with the
    device product and device comparison; rank and pivot columns are checked on the host."""
    import torch
    n = 65536
    w = n // 64
    t = torch.zeros((n, 2 * w), dtype=torch.int64, device="cuda")
    T = dev.DMat.from_torch(t, 2 * n)
    A = dev.DMat.random(n, n, 11)
    a_t = torch.zeros((n, w), dtype=torch.int64, device="cuda")
    Acopy = dev.DMat.from_torch(a_t, n)
    dev.add(A, dev.DMat.from_torch(torch.zeros((n, w), dtype=torch.int64, device="cuda"), n), Acopy)  # copy through XOR with 0
    t[:, :w] = a_t
    idx = torch.arange(n, device="cuda")
    t[idx, w + idx // 64] = torch.bitwise_left_shift(torch.ones(n, dtype=torch.int64, device="cuda"), idx % 64)
    torch.cuda.synchronize()
    rank, piv = dev.echelonize(T, full=True, ncols_limit=n)
    assert n - 70 < rank <= n and piv == sorted(piv) and len(set(piv)) == rank
    R = dev.DMat.wrap(t.data_ptr(), n, n, 2 * w, keep=t)
    E = dev.DMat.wrap(t.data_ptr() + 8 * w, n, n, 2 * w, keep=t)
    prod = dev.mul(E, A, algo="auto")
    assert dev.equal(prod, R)


ELIM = sorted(__import__("glob").glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "elim", "*.npz")))


@pytest.mark.parametrize("path", ELIM, ids=[os.path.basename(p)[:-4] for p in ELIM])
def test_elimination_fixtures(pkg, path):
    """The committed known answers (independent numpy elimination, tests/golden/make_golden_elim.py), through the C ABI."""
    d = np.load(path)
    name = os.path.basename(path)
    if name.startswith("rref"):
        m, n = (int(x) for x in d["shape"])
        got, rank = _host_rref(pkg, d["a"], n)
        assert rank == len(d["pivots"]) and np.array_equal(got, d["rref"])
        assert pkg.BinMatrix.from_words(d["a"], n).rank() == rank
    elif name.startswith("inverse"):
        n = int(d["shape"][0])
        assert np.array_equal(pkg.BinMatrix.from_words(d["a"], n).inverted().to_words(), d["inv"])
    else:
        m, n, k = (int(x) for x in d["shape"])
        B = pkg.BinMatrix.from_words(d["b"], k)
        assert pkg.solve_left(pkg.BinMatrix.from_words(d["a"], n), B) is True
        assert np.array_equal(B.to_words(), d["x"])
        B2 = pkg.BinMatrix.from_words(d["b_inconsistent"], k)
        assert pkg.solve_left(pkg.BinMatrix.from_words(d["a"], n), B2) is (not bool(d["inconsistent"][0]))


def test_device_operators(pkg, dev):
    """DMat mirrors the friendly operators on device-resident operands: chained work without PCIe round trips."""
    n = 3000
    a, b = g.random_words(n, n, 5), g.random_words(n, n, 6)
    A, B = dev.DMat.from_words(a, n), dev.DMat.from_words(b, n)
    ab = g.o_mul_m4rm(a, b, n, n, n)
    assert np.array_equal((A * B).to_words(), ab)
    assert np.array_equal((A + B).to_words(), a ^ b)
    assert (A * B).transposed() == B.transposed() * A.transposed()
    assert A.clone() == A and not (A == B)
    assert A.rank() == g.o_echelonize(a, n, n)[1] and np.array_equal(A.to_words(), a)
    inv = dev.DMat.from_words(_invertible(500, 77), 500)
    ident = dev.DMat.from_words(g.bits_to_words(np.eye(500, dtype=np.uint8)), 500)
    assert inv * inv.inverted() == ident


def test_concurrent_host_threads_elimination(pkg):
    """BinMatrix is Send + Sync upstream (binary_matrix.rs:38-39): several host threads run rank / inverse / products on
    shared inputs at once through the C ABI (ctypes drops the GIL); every call owns its stream and buffers."""
    import threading
    n = 1500
    a = _invertible(n, 5)
    low = _low_rank(1200, 1700, 333, 6)
    A, Lo = pkg.BinMatrix.from_words(a, n), pkg.BinMatrix.from_words(low, 1700)
    inv_ref = g.o_inverse(a, n)
    rank_ref = g.o_echelonize(low, 1200, 1700)[1]
    rref_ref = g.o_echelonize(low, 1200, 1700, full=True)[0]
    ident = pkg.BinMatrix.identity(n)
    errors = []

    def worker(k):
        try:
            for it in range(4):
                what = (k + it) % 4
                if what == 0:
                    ok = np.array_equal(A.inverted().to_words(), inv_ref)
                elif what == 1:
                    ok = Lo.rank() == rank_ref
                elif what == 2:
                    c = Lo.clone()
                    ok = c.echelonize(full=True) == rank_ref and np.array_equal(c.to_words(), rref_ref)
                else:
                    ok = (A * A.inverted()) == ident
                if not ok:
                    errors.append((k, it, what))
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("small_work", ["0", str(1 << 20)], ids=["device", "host-dispatch"])
def test_invert_naive(pkg, monkeypatch, small_work):
    """mzd_invert_naive (mzd.rs:214-218) against the oracle's inverse, through the device elimination and through the host
    routine of the size dispatch (the shipped default); with a caller-supplied identity, a preallocated result, and a
    singular input (NULL)."""
    import ctypes
    monkeypatch.setenv("M4RI_HIP_HOST_SMALL_WORK", small_work)
    L = pkg._lib.lib()
    done = 0
    for n, seed in [(1, 1), (7, 2), (64, 3), (65, 4), (100, 5), (130, 6), (700, 7)]:
        for s2 in range(seed, seed + 40):
            a = g.random_words(n, n, 1000 * n + s2)
            inv = g.o_inverse(a, n)
            if inv is not None:
                break
        else:
            continue
        A = pkg.BinMatrix.from_words(a, n)
        r = L.mzd_invert_naive(None, A.mzd, None)
        assert r
        R = pkg.BinMatrix(r)
        assert np.array_equal(R.to_words(), inv), n
        I = pkg.BinMatrix.identity(n)
        out = pkg.BinMatrix.from_words(g.random_words(n, n, 9), n)
        r2 = L.mzd_invert_naive(out.mzd, A.mzd, I.mzd)
        assert r2 and ctypes.addressof(r2.contents) == ctypes.addressof(out.mzd.contents)
        assert np.array_equal(out.to_words(), inv)
        done += 1
    assert done >= 4
    z = np.zeros((20, 1), dtype=np.uint64)
    z[3, 0] = 5
    Z = pkg.BinMatrix.from_words(z, 20)
    assert not L.mzd_invert_naive(None, Z.mzd, None)
    # rank n - 1 (a product of an n x (n - 1) and an (n - 1) x n matrix): [A | I] still has rank n, so only the reduced form's
    # left block tells -- NULL on both the device path and the host path of the size dispatch
    for n in (33, 200):
        low = g.o_mul_naive(g.random_words(n, n - 1, 70 + n), g.random_words(n - 1, n, 71 + n), n, n - 1, n)
        assert g.o_inverse(low, n) is None
        assert not L.mzd_invert_naive(None, pkg.BinMatrix.from_words(low, n).mzd, None)


_FAULT_SCRIPT = r"""
import sys, time
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import m4ri_rust_amd as pkg
from m4ri_rust_amd import device, _lib
n = 4096
M = device.DMat.random(n, n, 5)
t0 = time.time()
try:
    device.echelonize(M, full=True)
except _lib.HipError as e:
    print("FAILED-AS-DOCUMENTED %%.1f s: %%s" %% (time.time() - t0, e))
    # the library is usable afterwards: a product on the same stream still matches the oracle
    import gf2util as g
    a, b = g.random_words(300, 200, 1), g.random_words(200, 100, 2)
    c = device.mul(device.DMat.from_words(a, 200), device.DMat.from_words(b, 100)).to_words()
    assert np.array_equal(c, g.o_mul_m4rm(a, b, 300, 200, 100))
    print("STILL-USABLE")
    sys.exit(0)
print("NO-ERROR after %%.1f s" %% (time.time() - t0))
sys.exit(3)
"""


def test_lookahead_failure_is_reported_not_hung(pkg):
    """The in-launch look-ahead (gf2_elim.hip) has one workgroup wait for counters the update workgroups of the SAME launch raise.  If
    they never arrive -- M4RI_HIP_ELIM_FAULT=1 makes update workgroup 0 skip its raises -- the wait is bounded (1 s of s_memrealtime),
    the look-ahead workgroup leaves WITHOUT searching on, publishing or resetting the counters, every later launch of the chain sees
    the error flag and touches nothing, and the host reports the failure: an error return, not a hang and not a corrupted matrix
    handed back as a result.  In a child process: the library reads the hook once."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, M4RI_HIP_ELIM_FAULT="1")
    r = subprocess.run([sys.executable, "-c", _FAULT_SCRIPT % {"root": root}], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "FAILED-AS-DOCUMENTED" in r.stdout and "look-ahead" in r.stdout and "STILL-USABLE" in r.stdout, r.stdout


@pytest.mark.parametrize("knobs", [{"M4RI_HIP_ELIM_LOOKAHEAD": "0"}, {"M4RI_HIP_ELIM_SPECULATE": "0"},
                                   {"M4RI_HIP_ELIM_LOOKAHEAD": "0", "M4RI_HIP_ELIM_SPECULATE": "0"}],
                         ids=["two-launch steps", "waiting products", "both"])
def test_elimination_switches_of_the_shipped_library(pkg, knobs):
    """M4RI_HIP_ELIM_LOOKAHEAD=0 / M4RI_HIP_ELIM_SPECULATE=0 are read by the shipped library (a part on which the look-ahead launch
    cannot be resident as a whole must be able to turn it off): same reduced echelon form either way, checked in a child process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
              "import numpy as np, gf2util as g, m4ri_rust_amd as pkg\n"
              "for (m, n, r) in ((3000, 2500, 1800), (5000, 4200, 4200), (2048, 4096, 2048)):\n"
              "    a = g.o_mul_naive(g.random_words(m, r, 7), g.random_words(r, n, 8), m, r, n)\n"
              "    M = pkg.BinMatrix.from_words(a, n)\n"
              "    rank = M.echelonize(full=True)\n"
              "    ref, orank, _ = g.o_echelonize(a, m, n, full=True)\n"
              "    assert rank == orank and np.array_equal(M.to_words(), ref), (m, n, r)\n"
              "print('OK')\n") % (root, root)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=dict(os.environ, **knobs), timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
