"""Randomised shape fuzz of the product and elimination entry points against the oracle (development tool).
    python tools/fuzz_shapes.py [count] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gf2util as g
import m4ri_rust_amd as pkg
from m4ri_rust_amd import device as dev

count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def dim(hi):
    return int(np.exp(rng.uniform(0, np.log(hi)))) or 1


bad = 0
for it in range(count):
    hi = int(os.environ.get("FUZZ_MAX_DIM", "6000"))
    m, l, n = dim(hi), dim(hi), dim(hi)
    if rng.random() < 0.2:
        m = int(rng.choice([2048, 4096, 5000, 70000]))
        n = dim(256)
    a, b = g.random_words(m, l, 2 * it + 1), g.random_words(l, n, 2 * it + 2)
    ref = g.o_mul_m4rm(a, b, m, l, n)
    A, B = dev.DMat.from_words(a, l), dev.DMat.from_words(b, n)
    for algo in ("auto", "m4rm", "naive"):
        got = dev.mul(A, B, algo=algo).to_words()
        if not np.array_equal(got, ref):
            bad += 1
            print("MISMATCH product", m, l, n, algo, flush=True)
    if it % 4 == 0 and m * n < 4_000_000:
        M = pkg.BinMatrix.from_words(a, l)
        rank = M.echelonize(full=True)
        red, orank, _ = g.o_echelonize(a, m, l, full=True)
        if rank != orank or not np.array_equal(M.to_words(), red):
            bad += 1
            print("MISMATCH rref", m, l, flush=True)
    if it % 25 == 0:
        print("..", it, "done", flush=True)
print("fuzz finished:", count, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
