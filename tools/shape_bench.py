#!/usr/bin/env python3
"""Device-resident timing of arbitrary product shapes (development tool).

    python tools/shape_bench.py m,l,n[,algo[,levels[,reps[,acc[,lfull]]]]] ...

acc = 1: C ^= A*B (accumulate); lfull > l: A is the column window [0, l) of an m x lfull matrix (row stride lfull / 64 words) -- the
operand shapes of a host product pipelined over slabs of the inner dimension.

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
        acc = bool(int(f[6])) if len(f) > 6 else False
        lfull = int(f[7]) if len(f) > 7 else l
        B, C = dev.DMat.random(l, n, 2), dev.DMat(m, n)
        if lfull > l:
            Afull = dev.DMat.random(m, lfull, 1)
            A = dev.DMat.wrap(Afull.s.data, m, l, Afull.ld, keep=Afull)
        else:
            A = dev.DMat.random(m, l, 1)
        if acc:
            C.fill_random(3)
        _mul = dev.mul
        dev_mul = (lambda a, b, c, algo="auto", param=0: _mul(a, b, C=c, accumulate=True, algo=algo, param=param)) if acc else _mul
        for _ in range(2):
            dev_mul(A, B, C, algo=algo, param=levels)
        torch.cuda.synchronize()
        # ~30 ms of untimed products: the clocks of an idle GPU need more than a couple of milliseconds of work to come up
        # (a 35-us product measures 8 % slow after two warm-up products)
        t0 = time.perf_counter()
        dev_mul(A, B, C, algo=algo, param=levels)
        torch.cuda.synchronize()
        for _ in range(min(2000, int(0.03 / max(time.perf_counter() - t0, 1e-6)))):
            dev_mul(A, B, C, algo=algo, param=levels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            dev_mul(A, B, C, algo=algo, param=levels)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        lv = dev._lib.lib().gf2_strassen_levels(m, l, n, dev.ALGOS[algo], levels)
        print("%6d x %6d x %6d  %-8s levels %d%s%s: %9.3f ms  %.3e bit-ops/s" % (m, l, n, algo, lv, " accumulate" if acc else "",
              " (A = window of %d columns)" % lfull if lfull > l else "", dt * 1e3, 2.0 * m * l * n / dt), flush=True)
        del A, B, C


if __name__ == "__main__":
    main()
