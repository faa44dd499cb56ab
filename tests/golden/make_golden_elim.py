#!/usr/bin/env python3
"""Generates tests/golden/elim/*.npz: known answers for the elimination entry points.

The reference holds no test for rank / echelonize / inverted / solve_left and cannot be run here (see
make_golden.py), so these vectors come from an INDEPENDENT elimination: numpy row operations on the unpacked 0/1
matrix, written here, sharing no code with oracle/ or the HIP kernels.  Reduced row echelon forms, inverses and
solutions with free variables 0 are unique, so any correct implementation reproduces them bit for bit.

Inputs are reproducible (splitmix64 words, tests/gf2util.py); rank-deficient inputs are products of thin
random factors (numpy integer product mod 2, as in make_golden.py).

Run:  python tests/golden/make_golden_elim.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gf2util as g  # noqa: E402

OUT = os.path.join(HERE, "elim")


def rref_bits(bits, limit=None):
    """Gauss-Jordan on a 0/1 uint8 array; pivots only among the first `limit` columns."""
    a = bits.copy()
    m, n = a.shape
    limit = n if limit is None else limit
    r, piv = 0, []
    for c in range(limit):
        if r == m:
            break
        nz = np.nonzero(a[r:, c])[0]
        if nz.size == 0:
            continue
        p = r + int(nz[0])
        if p != r:
            a[[r, p]] = a[[p, r]]
        rows = np.nonzero(a[:, c])[0]
        rows = rows[rows != r]
        a[rows] ^= a[r]
        piv.append(c)
        r += 1
    return a, piv


def product_bits(x, y):
    return ((x.astype(np.int64) @ y.astype(np.int64)) & 1).astype(np.uint8)


def rand_bits(m, n, seed):
    return g.words_to_bits(g.random_words(m, n, seed), n)


def main():
    os.makedirs(OUT, exist_ok=True)
    # (name, rows, cols, rank bound or None)
    cases = [("rref_1x1", 1, 1, None), ("rref_64x64", 64, 64, None), ("rref_100x100", 100, 100, None),
             ("rref_70x130", 70, 130, None), ("rref_130x70", 130, 70, None), ("rref_257x513", 257, 513, None),
             ("rref_lowrank_300x400_r37", 300, 400, 37), ("rref_lowrank_640x640_r64", 640, 640, 64),
             ("rref_lowrank_500x200_r129", 500, 200, 129)]
    for name, m, n, r in cases:
        a = rand_bits(m, n, 1000 + m + n) if r is None else product_bits(rand_bits(m, r, 2000 + r), rand_bits(r, n, 3000 + r))
        red, piv = rref_bits(a)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), a=g.bits_to_words(a), rref=g.bits_to_words(red),
                            pivots=np.array(piv, dtype=np.int32), shape=np.array([m, n], dtype=np.int32))
    # inverses: unit lower x unit upper triangular
    for n in (1, 65, 200):
        lo = np.tril(rand_bits(n, n, 4000 + n), -1) | np.eye(n, dtype=np.uint8)
        up = np.triu(rand_bits(n, n, 5000 + n), 1) | np.eye(n, dtype=np.uint8)
        a = product_bits(lo, up)
        red, piv = rref_bits(np.concatenate([a, np.eye(n, dtype=np.uint8)], axis=1), limit=n)
        assert len(piv) == n
        inv = red[:, n:]
        assert np.array_equal(product_bits(a, inv), np.eye(n, dtype=np.uint8))
        np.savez_compressed(os.path.join(OUT, "inverse_%d.npz" % n), a=g.bits_to_words(a), inv=g.bits_to_words(inv),
                            shape=np.array([n, n], dtype=np.int32))
    # systems A X = B: consistent (B = A X0), free variables of the answer are 0; and one inconsistent system
    for name, m, n, k, r in (("solve_120x80x33", 120, 80, 33, None), ("solve_lowrank_200x150x70_r40", 200, 150, 70, 40)):
        a = rand_bits(m, n, 6000 + m) if r is None else product_bits(rand_bits(m, r, 6100 + r), rand_bits(r, n, 6200 + r))
        b = product_bits(a, rand_bits(n, k, 6300 + k))
        red, piv = rref_bits(np.concatenate([a, b], axis=1), limit=n)
        rank = len(piv)
        assert not red[rank:, n:].any()
        x = np.zeros((m, k), dtype=np.uint8)  # B has m >= n rows; rows n.. stay 0
        x[piv] = red[:rank, n:]
        assert np.array_equal(product_bits(a, x[:n]), b)
        bad = b.copy()
        bad[m - 1, 0] ^= 1
        red2, piv2 = rref_bits(np.concatenate([a, bad], axis=1), limit=n)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), a=g.bits_to_words(a), b=g.bits_to_words(b), x=g.bits_to_words(x),
                            b_inconsistent=g.bits_to_words(bad), inconsistent=np.array([int(red2[len(piv2):, n:].any())]),
                            shape=np.array([m, n, k], dtype=np.int32))
    print("wrote", len(os.listdir(OUT)), "files to", OUT)


if __name__ == "__main__":
    main()
