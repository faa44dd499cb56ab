#!/usr/bin/env python3
"""One (warm) mzd_mul on host matrices of size argv[1] (default 32768), for timeline traces:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d out -- python3 tools/host_one.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import m4ri_rust_amd as pkg
L = pkg._lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
A, B = pkg.BinMatrix.random(n, n), pkg.BinMatrix.random(n, n)
C = pkg.BinMatrix.zero(n, n)
for _ in range(3):
    L.mzd_mul(C.mzd, A.mzd, B.mzd, 0)
t0 = time.perf_counter()
L.mzd_mul(C.mzd, A.mzd, B.mzd, 0)
print("last call %.2f ms" % ((time.perf_counter() - t0) * 1e3))
