"""A few warm `&A * &v` (2^20 x 256, A uploaded by every call unless argv[1] == "cached") for a rocprofv3 timeline (development tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import m4ri_rust_amd as pkg
L = pkg._lib.lib()
m, l = 1 << 20, 256
A = pkg.BinMatrix.random(m, l)
vrow = pkg.BinMatrix.random(1, l)
if len(sys.argv) > 1 and sys.argv[1] == "cached":
    L.gf2_mzd_cache_on_device(A.mzd)
for _ in range(4):
    vt = L.mzd_transpose(None, vrow.mzd)
    r = L.mzd_mul_naive(None, A.mzd, vt)
    rt = L.mzd_transpose(None, r)
    L.mzd_free(vt), L.mzd_free(r), L.mzd_free(rt)
