#!/usr/bin/env python3
"""Generates tests/golden/digests_large.json: sha256(C) for the BASELINE.json configurations at their FULL size.

Which generator produced which digest (BASELINE.md section 3 parity gate):

* lpn_1048576x256x{1,64,128,256} (BASELINE config 5, 2^20 x 256 times 256 x V): the INDEPENDENT numpy product of
  make_golden.py (float32 matmul of the unpacked bits, mod 2; exact: sums <= 256), computed in row chunks.
  Shares no code with oracle/ or the kernels.
* sq_32768 and sq_65536 (BASELINE configs 3 and 4): oracle_mul_fast (oracle/gf2_oracle.c: single-thread M4RM k=8 +
  Strassen-Winograd), because the numpy product of 65536^3 would take hours.  Each is cross-checked here on a sample
  of rows (first, last, and a seeded random subset) against (a) oracle_mul_m4rm -- plain Four Russians, no Strassen,
  Gray-code tables: a different algorithm of the oracle -- and (b) the numpy product restricted to those rows
  (independent of the oracle), and the script refuses to write a digest whose sample disagrees.

Inputs: word t of A = splitmix64(1, t), of B = splitmix64(2, t) (tests/gf2util.random_words; the device generator
gf2_dmat_fill_random yields the same words).  Digest = sha256 of the row-major uint64 words of C.

Run (about 10 minutes and 12 GiB here):  python tests/golden/make_golden_large.py
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gf2util as g  # noqa: E402

SEED_A, SEED_B = 1, 2


def numpy_rows(a_rows, b, l, n, lchunk=4096):
    """Independent product of a few rows: float32 matmul of unpacked bits, chunked over the inner dimension so that
    every partial sum stays exactly representable (<= lchunk < 2^24)."""
    r = a_rows.shape[0]
    acc = np.zeros((r, n), dtype=np.int64)
    A = g.words_to_bits(a_rows, l)
    for l0 in range(0, l, lchunk):
        l1 = min(l, l0 + lchunk)
        Bc = g.words_to_bits(b[l0:l1], n).astype(np.float32)
        acc += np.rint(A[:, l0:l1].astype(np.float32) @ Bc).astype(np.int64)
    return g.bits_to_words((acc & 1).astype(np.uint8))


def lpn(v):
    m, l = 1 << 20, 256
    b = g.random_words(l, v, SEED_B)
    h = hashlib.sha256()
    step = 1 << 16
    wa = g.width(l)
    for r0 in range(0, m, step):  # rows [r0, r0 + step) of the seeded A: words r0 * wa ... of the stream
        t = np.arange(r0 * wa, (r0 + step) * wa, dtype=np.uint64)
        a = g.splitmix64(SEED_A, t).reshape(step, wa)
        h.update(g.numpy_mul(np.ascontiguousarray(a), b, step, l, v).tobytes())
    return {"m": m, "l": l, "n": v, "seed_a": SEED_A, "seed_b": SEED_B, "sha256_c": h.hexdigest(),
            "generator": "numpy float32 matmul of unpacked bits mod 2, row chunks of 65536 (independent of oracle/ and the kernels)"}


def square(n, nsample=24):
    a, b = g.random_words(n, n, SEED_A), g.random_words(n, n, SEED_B)
    t0 = time.time()
    c = g.o_mul_fast(a, b, n, n, n)
    dt = time.time() - t0
    rng = np.random.default_rng(n)
    rows = np.unique(np.concatenate([[0, n - 1, n // 2 - 1, n // 2], rng.integers(0, n, nsample)]))
    a_rows = np.ascontiguousarray(a[rows])
    via_m4rm = g.o_mul_m4rm(a_rows, b, len(rows), n, n)
    via_numpy = numpy_rows(a_rows, b, n, n)
    if not (np.array_equal(via_m4rm, c[rows]) and np.array_equal(via_numpy, c[rows])):
        raise SystemExit("sq_%d: oracle_mul_fast disagrees with the row sample -- no digest written" % n)
    return {"m": n, "l": n, "n": n, "seed_a": SEED_A, "seed_b": SEED_B, "sha256_c": hashlib.sha256(c.tobytes()).hexdigest(),
            "generator": "oracle_mul_fast (M4RM k=8 + Strassen-Winograd, %.0f s); %d rows cross-checked against oracle_mul_m4rm "
                         "and the independent numpy product" % (dt, len(rows)),
            "sample_rows": [int(r) for r in rows],
            "sample_rows_sha256": hashlib.sha256(np.ascontiguousarray(c[rows]).tobytes()).hexdigest()}


def main():
    out = os.path.join(HERE, "digests_large.json")
    dig = {}
    if os.path.exists(out):
        with open(out) as f:
            dig = json.load(f)
    want = sys.argv[1:] or ["lpn_1048576x256x1", "lpn_1048576x256x64", "lpn_1048576x256x128", "lpn_1048576x256x256", "sq_32768", "sq_65536"]
    for name in want:
        t0 = time.time()
        dig[name] = lpn(int(name.rsplit("x", 1)[1])) if name.startswith("lpn_") else square(int(name[3:]))
        print(name, dig[name]["sha256_c"][:16], "%.0f s" % (time.time() - t0), flush=True)
        with open(out, "w") as f:
            json.dump(dig, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
