#!/usr/bin/env python3
"""Generates tests/golden/*.npz and digests.json.

The reference (thomwiggers/m4ri-rust) cannot be built or imported in this environment: its
arithmetic lives in the M4RI C library, an un-checked-out submodule (m4ri-sys/vendor/m4ri is
empty), and there is no Rust toolchain.  Its own tests pin the multiply only on identity
products (m4ri-rust/src/friendly/binary_matrix.rs:662-686).  These vectors are therefore
produced by an INDEPENDENT bit-level product -- numpy integer matmul of the unpacked 0/1
matrices, reduced mod 2 -- which shares no code with oracle/ or with the HIP kernels, on the
shapes of the reference's bench file (m4ri-rust/benches/binary_matrix.rs:30-76) plus ragged
sizes around the 64-bit word boundary.

Inputs are reproducible: word t of matrix X = splitmix64(seed_X, t) (tests/gf2util.py), excess
bits of each row's last word cleared (the M4RI convention the reference relies on,
binary_matrix.rs:151-155).  Small cases store A, B and C words; large cases store sha256(C).

Run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gf2util as g  # noqa: E402

# (name, m, l, n)  -- bench shapes of the reference first
FULL = [
    ("bench_10x10x10", 10, 10, 10),
    ("bench_100x10x10", 100, 10, 10),
    ("bench_100x10x100", 100, 10, 100),
    ("bench_1000x64x1000", 1000, 64, 1000),
    ("bench_1000x10x1000", 1000, 10, 1000),
    ("bench_1000x64x1", 1000, 64, 1),
    ("bench_1x64x1000", 1, 64, 1000),
    ("bench_10x1000x10", 10, 1000, 10),
    ("bench_vecmat_1x1000x64", 1, 1000, 64),
    ("bench_matvec_64x1000x1", 64, 1000, 1),
    ("bench_mulslice_34x128x1", 34, 128, 1),
    ("ragged_1x1x1", 1, 1, 1),
    ("ragged_63x64x65", 63, 64, 65),
    ("ragged_65x127x129", 65, 127, 129),
    ("ragged_127x63x1", 127, 63, 1),
    ("ragged_129x65x63", 129, 65, 63),
    ("ragged_257x513x300", 257, 513, 300),
    ("ragged_1000x1000x1000", 1000, 1000, 1000),
    ("lpn_4096x256x1", 4096, 256, 1),
    ("lpn_4096x256x64", 4096, 256, 64),
]
DIGEST_ONLY = [
    ("sq_1024", 1024, 1024, 1024),
    ("sq_2048", 2048, 2048, 2048),
    ("sq_4096", 4096, 4096, 4096),
    ("rect_3000x2100x2500", 3000, 2100, 2500),
    ("lpn_65536x256x256", 65536, 256, 256),
]
SEED_A, SEED_B = 1, 2


def main():
    digests = {}
    for name, m, l, n in FULL + DIGEST_ONLY:
        a = g.random_words(m, l, SEED_A)
        b = g.random_words(l, n, SEED_B)
        c = g.numpy_mul(a, b, m, l, n)
        digests[name] = {"m": m, "l": l, "n": n, "seed_a": SEED_A, "seed_b": SEED_B,
                         "sha256_c": hashlib.sha256(c.tobytes()).hexdigest()}
        if (name, m, l, n) in FULL:
            np.savez_compressed(os.path.join(HERE, name + ".npz"), a=a, b=b, c=c,
                                dims=np.array([m, l, n], dtype=np.int64))
        print(name, digests[name]["sha256_c"][:16])
    # the reference's own known answers (binary_matrix.rs:662-686): I*I = I, I*1 = 1, 1*I = 1
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
