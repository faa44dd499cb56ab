"""Shape fuzz of the HOST entry points (round 5: result side copies, pooled row-pointer arrays, the zero-copy vector product, the
schedules of large host products): random shapes through mzd_mul / mzd_mul_m4rm / mzd_mul_naive on host mzd_t -- NULL and
preallocated destinations, A and / or B cached on the device, every schedule of the host pipeline for the large ones -- against the
oracle, followed by mzd_transpose(NULL, product) (served from the side copy where the product carries one) against the oracle's
transposition, once more after a library write to the product (mzd_add), and a product straight after the matrix was freed and another
of the same shape allocated (pooled blocks must not bring stale side copies or row pointers along).
    python tools/fuzz_host.py [count] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("M4RI_HIP_HOST_SMALL_WORK", "0")
import gf2util as g
import m4ri_rust_amd as pkg

L = pkg._lib.lib()
count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)


def logdim(lo, hi):
    return max(lo, int(np.exp(rng.uniform(np.log(lo), np.log(hi)))))


bad = 0
for it in range(count):
    kind = it % 5
    if kind == 0:    # the LPN shape and its neighbours: many rows, up to 256 bits, one to eight vectors
        m, l, n = logdim(100000, 1500000), int(rng.integers(65, 257)), int(rng.integers(1, 9))
    elif kind == 1:  # thin products in general
        m, l, n = logdim(20000, 600000), logdim(1, 1024), int(rng.integers(1, 257))
    elif kind == 2:  # large enough for the pipelined schedules (multiples the slabs accept, and not)
        m = int(rng.choice([16384, 16384 + 256, 20480, 32768]))
        l = int(rng.choice([16384, 32768, 16384 + 128]))
        n = int(rng.choice([16384, 16384 + 77, 8192 + 64, 24576]))
        os.environ["M4RI_HIP_HOST_PLAN"] = str(int(rng.integers(0, 13)))
    else:            # anything mid-sized
        m, l, n = logdim(1, 6000), logdim(1, 6000), logdim(1, 6000)
    a, b = g.random_words(m, l, 7 * it + 1), g.random_words(l, n, 7 * it + 2)
    A, B = pkg.BinMatrix.from_words(a, l), pkg.BinMatrix.from_words(b, n)
    ref = g.o_mul_fast(a, b, m, l, n) if kind == 2 else g.o_mul_m4rm(a, b, m, l, n, k=8)
    if rng.integers(0, 3) == 0:
        A.cache_on_device()
    if rng.integers(0, 4) == 0:
        B.cache_on_device()
    fns = [("mzd_mul", lambda c: L.mzd_mul(c, A.mzd, B.mzd, 0)), ("mzd_mul_m4rm", lambda c: L.mzd_mul_m4rm(c, A.mzd, B.mzd, 0)),
           ("mzd_mul_naive", lambda c: L.mzd_mul_naive(c, A.mzd, B.mzd))]
    name, fn = fns[int(rng.integers(0, 3))]
    R = pkg.BinMatrix(fn(None))
    if not np.array_equal(R.to_words(), ref):
        bad += 1
        print("MISMATCH", name, m, l, n, os.environ.get("M4RI_HIP_HOST_PLAN"), flush=True)
    T = R.transposed()
    if not np.array_equal(T.to_words(), g.o_transpose(ref, m, n)):
        bad += 1
        print("MISMATCH transpose of the product", name, m, l, n, flush=True)
    y = g.random_words(m, n, 7 * it + 3)
    Y = pkg.BinMatrix.from_words(y, n)
    L.mzd_add(R.mzd, R.mzd, Y.mzd)
    if not np.array_equal(R.transposed().to_words(), g.o_transpose(ref ^ y, m, n)):
        bad += 1
        print("MISMATCH transpose after mzd_add", name, m, l, n, flush=True)
    Cp = pkg.BinMatrix.from_words(y, n)  # preallocated: overwritten
    if not fn(Cp.mzd) or not np.array_equal(Cp.to_words(), ref):
        bad += 1
        print("MISMATCH preallocated", name, m, l, n, flush=True)
    del R, T, Cp
    # pooled blocks: a second product of the same shape right after the first was freed, different operands
    a2 = g.random_words(m, l, 7 * it + 4)
    A2 = pkg.BinMatrix.from_words(a2, l)
    R2 = pkg.BinMatrix(L.mzd_mul_naive(None, A2.mzd, B.mzd))
    ref2 = g.o_mul_fast(a2, b, m, l, n) if kind == 2 else g.o_mul_m4rm(a2, b, m, l, n, k=8)
    if not np.array_equal(R2.to_words(), ref2) or not np.array_equal(R2.transposed().to_words(), g.o_transpose(ref2, m, n)):
        bad += 1
        print("MISMATCH second product / its transpose", m, l, n, flush=True)
    os.environ.pop("M4RI_HIP_HOST_PLAN", None)
    del A, B, A2, R2, Y
    if it % 10 == 9:
        print("..", it + 1, "done", flush=True)
print("fuzz_host finished: %d cases, %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
