"""A few warm mzd_mul on host mzd_t (n from argv, schedule from M4RI_HIP_HOST_PLAN) for a rocprofv3 timeline (development tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import m4ri_rust_amd as pkg
L = pkg._lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
A, B = pkg.BinMatrix.random(n, n), pkg.BinMatrix.random(n, n)
C = pkg.BinMatrix.zero(n, n)
for _ in range(3):
    L.mzd_mul(C.mzd, A.mzd, B.mzd, 0)
