"""Randomised shapes through the Strassen driver -- every level plan, padded and peeled shapes, accumulate form -- against the
plain tile kernel on the device and, on sampled rows, against the oracle (development tool).
    python tools/fuzz_strassen.py [count] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gf2util as g
from m4ri_rust_amd import device as dev

count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for it in range(count):
    base = int(rng.choice([1024, 2048, 4096, 8192]))
    def dim():
        k = int(rng.integers(1, 5))
        d = base * k
        r = rng.random()
        if r < 0.35:
            d += int(rng.integers(-200, 200))
        elif r < 0.5:
            d += int(rng.choice([-64, 64, 1, -1, 63, 65, 128]))
        return max(d, 300)
    m, l, n = dim(), dim(), dim()
    levels = int(rng.integers(0, 5))
    A, B = dev.DMat.random(m, l, 3 * it + 1), dev.DMat.random(l, n, 3 * it + 2)
    ref = dev.mul(A, B, algo="m4rm")
    acc = bool(rng.random() < 0.4)
    if acc:
        C = dev.DMat.random(m, n, 3 * it + 3)
        expect = dev.add(C, ref)
        dev.mul(A, B, C=C, accumulate=True, algo="strassen", param=levels)
    else:
        expect = ref
        C = dev.mul(A, B, algo="strassen", param=levels)
    ok = dev.equal(C, expect)
    if it % 6 == 0:  # and the plain kernel itself against the oracle on a few rows
        rows = sorted(set(int(x) for x in rng.integers(0, m, 5)) | {0, m - 1})
        a_rows = np.ascontiguousarray(g.random_words(m, l, 3 * it + 1)[rows])
        ok = ok and np.array_equal(ref.to_words()[rows], g.o_mul_m4rm(a_rows, g.random_words(l, n, 3 * it + 2), len(rows), l, n))
    if not ok:
        bad += 1
        print("MISMATCH", m, l, n, "levels", levels, "accumulate", acc, flush=True)
    if it % 10 == 0:
        print("..", it, "done", flush=True)
print("strassen fuzz finished:", count, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
