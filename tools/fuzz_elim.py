"""Randomised structure fuzz of the BLOCKED elimination path (more than 1024 rows, or too large for the one-workgroup kernel) against
the oracle (development tool; round 5: the stash, the two-row update, the in-lane insertion loop and their fallbacks all sit in this path).
    python tools/fuzz_elim.py [count] [seed] [tall]
("tall": 49,000-140,000 rows x 65-320 columns -- from 192 rows per update workgroup on the publication goes ahead of the update's end.)
Every case draws a shape, a column-block width and a recipe of planted structure: low rank, zero rows, rows repeated k times in a run,
equal column pairs, zero column bands, rows that are sums of others, a shuffled row order; reduced and upper form are both checked."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gf2util as g
import m4ri_rust_amd as pkg

count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
TALL = len(sys.argv) > 3 and sys.argv[3] == "tall"


def make(it):
    m = int(rng.integers(49000, 140000)) if TALL else int(rng.integers(1025, 5200))
    n = int(np.exp(rng.uniform(np.log(65), np.log(320 if TALL else 5000))))
    recipe = []
    if rng.random() < 0.5:
        r = int(np.exp(rng.uniform(0, np.log(min(m, n)))))
        a = g.o_mul_m4rm(g.random_words(m, r, 11 * it), g.random_words(r, n, 11 * it + 1), m, r, n)
        recipe.append("rank<=%d" % r)
    else:
        a = g.random_words(m, n, 11 * it + 2)
    bits = g.words_to_bits(a, n)
    if rng.random() < 0.4:
        k = int(rng.integers(2, 12))
        bits = np.repeat(bits[: (m + k - 1) // k], k, axis=0)[:m]
        recipe.append("rows x%d" % k)
    if rng.random() < 0.4:
        p = rng.uniform(0.1, 0.9)
        bits[rng.random(m) < p] = 0
        recipe.append("zero rows %.1f" % p)
    if rng.random() < 0.3 and n >= 4:
        bits = np.repeat(bits[:, : (n + 1) // 2], 2, axis=1)[:, :n]
        recipe.append("column pairs")
    if rng.random() < 0.4:
        for _ in range(int(rng.integers(1, 4))):
            c0 = int(rng.integers(0, n)); c1 = min(n, c0 + int(rng.integers(1, 300)))
            bits[:, c0:c1] = 0
        recipe.append("zero bands")
    if rng.random() < 0.3:
        q = m // 3
        bits[2 * q: 3 * q] = bits[:q] ^ bits[q: 2 * q]
        recipe.append("sums")
    if rng.random() < 0.3:
        bits = bits[rng.permutation(m)]
        recipe.append("shuffled")
    return m, n, np.ascontiguousarray(bits), recipe


bad = 0
for it in range(count):
    m, n, bits, recipe = make(it)
    bw = int(rng.choice([2, 5, 32]))
    os.environ["M4RI_HIP_ELIM_BLOCK_WORDS"] = str(bw)
    a = g.bits_to_words(bits)
    red, orank, piv = g.o_echelonize(a, m, n, full=True)
    for full in (True, False):
        M = pkg.BinMatrix.from_words(a, n)
        rank = M.echelonize(full=full)
        ok = rank == orank
        if full:
            ok = ok and np.array_equal(M.to_words(), red)
        else:
            ok = ok and np.array_equal(g.o_echelonize(M.to_words(), m, n, full=True)[0], red)
        if not ok:
            bad += 1
            print("MISMATCH", m, n, "block words", bw, "full" if full else "upper", recipe, "rank", rank, orank, flush=True)
    if it % 20 == 0:
        print("..", it, "done (last: %d x %d, rank %d, %s)" % (m, n, orank, ", ".join(recipe) or "random"), flush=True)
print("fuzz_elim finished:", count, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
