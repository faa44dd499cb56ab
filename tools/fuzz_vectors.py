"""Shape fuzz aimed at the products with at most 64 columns (matrix x vector, blocks of vectors; mul_slice / `&A * &v` / `&v * &A`):
rows and inner dimensions log-uniform up to 200000 / 300000 (at most 2^31 bits of A), every selector, accumulate, the
pre-transposed entry, few-row products against narrow and wide B.      python tools/fuzz_vectors.py [count] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gf2util as g
from m4ri_rust_amd import device as dev

count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)


def logdim(hi):
    return max(1, int(np.exp(rng.uniform(0, np.log(hi)))))


bad = 0
for it in range(count):
    if it % 4 == 3:  # few rows times a matrix (the v*A kernel)
        m, l, n = int(rng.integers(1, 9)), logdim(100000), logdim(20000)
    else:
        m, l, n = logdim(200000), logdim(300000), int(rng.integers(1, 65))
        while m * l > 1 << 31:
            m = max(1, m // 2)
    a, b = g.random_words(m, l, 5 * it + 1), g.random_words(l, n, 5 * it + 2)
    ref = g.o_mul_m4rm(a, b, m, l, n, k=8)
    A, B = dev.DMat.from_words(a, l), dev.DMat.from_words(b, n)
    for algo in ("naive", "auto", "m4rm"):
        if not np.array_equal(dev.mul(A, B, algo=algo).to_words(), ref):
            bad += 1
            print("MISMATCH", m, l, n, algo, flush=True)
    c0 = g.random_words(m, n, 5 * it + 3)
    for algo in ("naive", "auto"):
        C = dev.DMat.from_words(c0, n)
        dev.mul(A, B, C, accumulate=True, algo=algo)
        if not np.array_equal(C.to_words(), c0 ^ ref):
            bad += 1
            print("MISMATCH accumulate", m, l, n, algo, flush=True)
    if n <= 64 and it % 3 == 0:
        if not np.array_equal(dev.mul_nt(A, dev.transpose(B)).to_words(), ref):
            bad += 1
            print("MISMATCH nt", m, l, n, flush=True)
    if it % 25 == 24:
        print("..", it + 1, "done", flush=True)
print("fuzz_vectors finished: %d cases, %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
