"""Shape fuzz aimed at the row-band plans: rows a little above a multiple of the tile height, many column tiles, short inner
dimensions (so that the oracle stays cheap); plain products and two Strassen levels over ragged leaves; accumulate.
    python tools/fuzz_bands.py [count] [seed]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gf2util as g
from m4ri_rust_amd import device as dev

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
L = dev._lib.lib()


def band_rows(m, l, n, batch, packed):
    out = (ctypes.c_longlong * 5)()
    L.gf2_tile_plan_band(m, l, n, batch, packed, out)
    return int(out[0])


bad = banded = 0
for it in range(count):
    R = int(rng.choice([2048, 4096]))
    m = R * int(rng.integers(1, 5)) + int(rng.integers(1, R // 6))
    l = int(rng.choice([rng.integers(40, 300), rng.integers(300, 1100), 64 * rng.integers(1, 40)]))
    n = int(rng.integers(12000, 72000))
    strassen = it % 4 == 3
    if strassen:  # two levels over leaves with ragged rows
        m = 4 * (R + 64 * int(rng.integers(1, R // 256)))
        l, n = 512 * int(rng.integers(2, 6)), 512 * int(rng.integers(8, 24))
        banded += band_rows(m >> 2, l >> 2, n >> 2, 49, 1) > 0
    else:
        banded += max(band_rows(m, l, n, 1, p) for p in (0, 1)) > 0
    a, b = g.random_words(m, l, 3 * it + 1), g.random_words(l, n, 3 * it + 2)
    rows = np.unique(np.concatenate([np.arange(0, m, 211), np.arange(max(0, m - 2100), m)]))
    ref = g.o_mul_m4rm(np.ascontiguousarray(a[rows]), b, len(rows), l, n, k=8)
    A, B = dev.DMat.from_words(a, l), dev.DMat.from_words(b, n)
    for algo, par in ([("strassen", 2)] if strassen else [("m4rm", 0), ("auto", 0)]):
        P = dev.mul(A, B, algo=algo, param=par)
        got = P.to_words()
        if not np.array_equal(got[rows], ref):
            bad += 1
            print("MISMATCH", m, l, n, algo, par, flush=True)
        if strassen and not dev.equal(P, dev.mul(A, B, algo="naive" if n <= 256 else "m4rm")):
            bad += 1
            print("MISMATCH against m4rm", m, l, n, flush=True)
    c0 = g.random_words(m, n, 3 * it + 3)
    C = dev.DMat.from_words(c0, n)
    dev.mul(A, B, C, accumulate=True, algo="strassen" if strassen else "m4rm", param=2 if strassen else 0)
    if not np.array_equal(C.to_words()[rows], c0[rows] ^ ref):
        bad += 1
        print("MISMATCH accumulate", m, l, n, flush=True)
    if it % 10 == 9:
        print("..", it + 1, "done,", banded, "with a band", flush=True)
print("fuzz_bands finished: %d cases (%d with a row band), %d mismatches" % (count, banded, bad))
sys.exit(1 if bad else 0)
