"""Timing of the device-resident elimination (gf2_echelonize_dev / gf2_inverse_dev) on random matrices.

    python tools/elim_bench.py [n ...]        # default 4096 8192 16384 32768 65536
Prints, per size: full reduced echelon form time, upper form time, inverse time, and the rate in "bit-ops/s"
counted as n^3 (reduced form of an n x n matrix: n^2/2 row additions of on average n/2... bits each, times 2 sides).
The CPU column is the oracle's textbook Gauss-Jordan on one core at n <= --cpu-max (a reported baseline only).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sizes", nargs="*", type=int, default=[4096, 8192, 16384, 32768, 65536])
    ap.add_argument("--cpu-max", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    if os.environ.get("AB_LIB"):  # A/B of two builds on one box: load this shared object instead
        import m4ri_rust_amd  # noqa
        from m4ri_rust_amd import _lib
        _lib.LIB_PATH = os.environ["AB_LIB"]
    import __graft_entry__ as ge
    ge.build()
    import m4ri_rust_amd as pkg  # noqa: F401
    from m4ri_rust_amd import device as dev
    import gf2util as g
    import torch

    for n in args.sizes:
        res = {}
        for name, full in (("rref", True), ("upper", False)):
            best = 1e9
            for _ in range(args.reps):
                A = dev.DMat.random(n, n, 5)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rank, _ = dev.echelonize(A, full=full)
                best = min(best, time.perf_counter() - t0)
                del A
            res[name] = best
        best = 1e9
        for _ in range(args.reps):
            A = dev.DMat.random(n, n, 5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            inv = dev.inverse(A)
            best = min(best, time.perf_counter() - t0)
            del A, inv
        res["inverse"] = best
        cpu = None
        if n <= args.cpu_max:
            a = g.random_words(n, n, 5)
            t0 = time.perf_counter()
            g.o_echelonize(a, n, n, full=True)
            cpu = time.perf_counter() - t0
        print("n=%6d rank=%6d  rref %9.2f ms (%.3e n^3/s)  upper %9.2f ms  inverse(singular stop or full) %9.2f ms  cpu-oracle rref %s"
              % (n, rank, res["rref"] * 1e3, n ** 3 / res["rref"], res["upper"] * 1e3, res["inverse"] * 1e3,
                 "%.2f s" % cpu if cpu else "-"), flush=True)


if __name__ == "__main__":
    main()
