#!/usr/bin/env python3
"""End-to-end time of the drop-in entry point on host mzd_t (upload A, B + product + download C), pinned blocks.
Development tool: the PCIe-inclusive rate quoted in DESIGN.md comes from here."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import m4ri_rust_amd as pkg
from m4ri_rust_amd import device
device.require_gpu()
L = pkg._lib.lib()
sizes = [int(x) for x in sys.argv[1:]] or [4096, 16384, 32768, 65536]
for n in sizes:
    A, B = pkg.BinMatrix.random(n, n), pkg.BinMatrix.random(n, n)
    c = L.mzd_mul(None, A.mzd, B.mzd, 0); L.mzd_free(c)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        c = L.mzd_mul(None, A.mzd, B.mzd, 0)
        L.mzd_free(c)
    dt = (time.perf_counter() - t0) / reps
    Cp = pkg.BinMatrix.zero(n, n)
    t0 = time.perf_counter()
    for _ in range(reps):
        L.mzd_mul(Cp.mzd, A.mzd, B.mzd, 0)
    dt2 = (time.perf_counter() - t0) / reps
    print(f"n={n}: mzd_mul(NULL,..) {dt*1e3:.1f} ms  preallocated C {dt2*1e3:.1f} ms  -> {2*n**3/dt2/1e12:.1f} Tbit-ops/s end to end; bytes moved {3*n*n/8/1e6:.0f} MB")

import time as _t
for n in ([] if len(sys.argv) > 1 else [16384, 65536]):
    A = pkg.BinMatrix.random(n, n)
    t0 = _t.perf_counter(); T = A.transposed(); first = _t.perf_counter() - t0
    del T  # the destination's pinned block goes back into the pool: the timed calls below reuse it (round 3 kept T alive, so
    #        every timed call pinned a fresh 512 MiB block -- ~100 ms of hipHostMalloc reported as if it were the transposition)
    ts = []
    for _ in range(3):
        t0 = _t.perf_counter(); T = A.transposed(); ts.append(_t.perf_counter() - t0); del T
    print(f"n={n}: mzd_transpose end to end {min(ts)*1e3:.1f} ms with a pooled destination block (first call of this process, incl. pinning: {first*1e3:.1f} ms)")
