#!/usr/bin/env python3
"""Device-resident timing of arbitrary product shapes (development tool).

    python tools/shape_bench.py m,l,n[,algo[,levels[,reps]]] ...

Prints one line per shape: milliseconds per product, bit-ops/s and the Strassen level count the library chose."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import m4ri_rust_amd  # noqa: F401
    from m4ri_rust_amd import _lib, device as dev
    if os.environ.get("AB_LIB"):  # A/B of two builds on one box: load this shared object instead
        _lib.LIB_PATH = os.environ["AB_LIB"]
    dev.require_gpu()
    for spec in sys.argv[1:]:
        f = spec.split(",")
        m, l, n = int(f[0]), int(f[1]), int(f[2])
        algo = f[3] if len(f) > 3 else "auto"
        levels = int(f[4]) if len(f) > 4 else 0
        reps = int(f[5]) if len(f) > 5 else 5
        A, B, C = dev.DMat.random(m, l, 1), dev.DMat.random(l, n, 2), dev.DMat(m, n)
        for _ in range(2):
            dev.mul(A, B, C, algo=algo, param=levels)
        torch.cuda.synchronize()
        # ~30 ms of untimed products: the clocks of an idle GPU need more than a couple of milliseconds of work to come up
        # (a 35-us product measures 8 % slow after two warm-up products)
        t0 = time.perf_counter()
        dev.mul(A, B, C, algo=algo, param=levels)
        torch.cuda.synchronize()
        for _ in range(min(2000, int(0.03 / max(time.perf_counter() - t0, 1e-6)))):
            dev.mul(A, B, C, algo=algo, param=levels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            dev.mul(A, B, C, algo=algo, param=levels)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        lv = dev._lib.lib().gf2_strassen_levels(m, l, n, dev.ALGOS[algo], levels)
        print("%6d x %6d x %6d  %-8s levels %d: %9.3f ms  %.3e bit-ops/s" % (m, l, n, algo, lv, dt * 1e3, 2.0 * m * l * n / dt),
              flush=True)
        del A, B, C


if __name__ == "__main__":
    main()
