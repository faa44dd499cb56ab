#!/usr/bin/env python3
"""Latency of the drop-in entry points on small host matrices (the shapes of the reference's own bench file,
m4ri-rust/benches/binary_matrix.rs:30-76): upload, product, download per call.  Development tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import m4ri_rust_amd as pkg
from m4ri_rust_amd import device
device.require_gpu()
L = pkg._lib.lib()
for (m, l, n) in ((10, 10, 10), (100, 10, 100), (1000, 64, 1000), (1000, 1000, 1000), (64, 1000, 1), (2048, 2048, 2048), (4096, 4096, 4096)):
    A, B = pkg.BinMatrix.random(m, l), pkg.BinMatrix.random(l, n)
    C = pkg.BinMatrix.zero(m, n)
    for fn, name in ((L.mzd_mul, "mzd_mul"), (L.mzd_mul_m4rm, "mzd_mul_m4rm")):
        fn(C.mzd, A.mzd, B.mzd, 0)
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(C.mzd, A.mzd, B.mzd, 0)
        dt = (time.perf_counter() - t0) / reps
        print(f"{m}x{l}x{n} {name}: {dt*1e6:.1f} us per call", flush=True)
    A2 = pkg.BinMatrix.random(m, l)
    t0 = time.perf_counter()
    for _ in range(50):
        r = A2.rank()
    print(f"{m}x{l} rank: {(time.perf_counter()-t0)/50*1e6:.1f} us per call", flush=True)
