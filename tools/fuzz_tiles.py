"""Shape fuzz aimed at the tile kernels' packed paths (m >= 2048, n >= 1024: A is packed, v6 / v7 run; ragged rows, inner
dimensions and column tiles; accumulate; explicit Strassen levels where the dimensions divide).  python tools/fuzz_tiles.py [count] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gf2util as g
from m4ri_rust_amd import device as dev

count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for it in range(count):
    m = int(rng.integers(2048, 9500))
    n = int(rng.integers(1024, 5200))
    l = int(rng.choice([rng.integers(1, 70), rng.integers(70, 600), rng.integers(600, 3300)]))
    if it % 5 == 0:  # divisible shapes: Strassen with packed leaves
        m, l, n = 256 * int(rng.integers(8, 30)), 512 * int(rng.integers(1, 6)), 512 * int(rng.integers(2, 9))
    a, b = g.random_words(m, l, 3 * it + 1), g.random_words(l, n, 3 * it + 2)
    ref = g.o_mul_m4rm(a, b, m, l, n)
    A, B = dev.DMat.from_words(a, l), dev.DMat.from_words(b, n)
    algos = [("m4rm", 0), ("auto", 0)] + ([("strassen", 2)] if it % 5 == 0 else [])
    for algo, par in algos:
        got = dev.mul(A, B, algo=algo, param=par).to_words()
        if not np.array_equal(got, ref):
            bad += 1
            print("MISMATCH", m, l, n, algo, par, flush=True)
    c0 = g.random_words(m, n, 3 * it + 3)
    C = dev.DMat.from_words(c0, n)
    dev.mul(A, B, C, accumulate=True, algo="m4rm")
    if not np.array_equal(C.to_words(), c0 ^ ref):
        bad += 1
        print("MISMATCH accumulate", m, l, n, flush=True)
    if it % 10 == 9:
        print("..", it + 1, "done", flush=True)
print("fuzz_tiles finished: %d cases, %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
