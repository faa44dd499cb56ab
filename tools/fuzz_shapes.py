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
    if it % 4 == 0 and m * l < 6_000_000:
        # elimination on the product's left operand, on a rank-deficient matrix (a product through a thin middle) and on a
        # matrix with planted structure (zero column blocks, repeated rows)
        cands = [a]
        r = dim(min(m, l))
        cands.append(g.o_mul_m4rm(g.random_words(m, r, 5 * it), g.random_words(r, l, 5 * it + 1), m, r, l))
        s_ = a.copy()
        if s_.shape[1] > 1:
            s_[:, int(rng.integers(0, s_.shape[1]))] = 0
        if m > 3:
            s_[m // 2:] = s_[: m - m // 2]
        cands.append(s_)
        for mat in cands:
            for full in (True, False):
                M = pkg.BinMatrix.from_words(mat, l)
                rank = M.echelonize(full=full)
                red, orank, piv = g.o_echelonize(mat, m, l, full=True)
                ok = rank == orank
                if full:
                    ok = ok and np.array_equal(M.to_words(), red)
                else:  # same row space: the reduced form of the result is the reduced form
                    ok = ok and np.array_equal(g.o_echelonize(M.to_words(), m, l, full=True)[0], red)
                if not ok:
                    bad += 1
                    print("MISMATCH echelon", m, l, "full" if full else "upper", flush=True)
        if m >= l:  # A X = B with B = A X0: the solver must return a solution (free variables 0) and report consistency
            k = dim(300)
            x0 = g.random_words(l, k, 7 * it)
            bmat = g.o_mul_m4rm(cands[1], x0, m, l, k)
            A2, B2 = pkg.BinMatrix.from_words(cands[1], l), pkg.BinMatrix.from_words(bmat, k)
            ok = pkg.solve_left(A2, B2)
            xref, okref = g.o_solve_left(cands[1], m, l, bmat, m, k)
            if not (ok and okref and np.array_equal(B2.to_words(), xref)):
                bad += 1
                print("MISMATCH solve", m, l, k, flush=True)
    if it % 25 == 0:
        print("..", it, "done", flush=True)
print("fuzz finished:", count, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
