"""mzd_transpose on host matrices (warm: the pinned pool holds the blocks), with and without the pipeline (development build: M4RI_HIP_TRANSPOSE_PIPELINE)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import m4ri_rust_amd as pkg
from m4ri_rust_amd import _lib
if os.environ.get("AB_LIB"):
    _lib.LIB_PATH = os.environ["AB_LIB"]
L = _lib.lib()
for (r, c) in [(65536, 65536), (32768, 65536), (65536, 32768)]:
    A = pkg.BinMatrix.random(r, c)
    T = pkg.BinMatrix.zero(c, r)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        L.mzd_transpose(T.mzd, A.mzd)
        best = min(best, time.perf_counter() - t0)
    print("pipeline=%s %d x %d: %.2f ms" % (os.environ.get("M4RI_HIP_TRANSPOSE_PIPELINE", "default"), r, c, best * 1e3), flush=True)
# the NULL-destination form (what `transposed()` calls): destination from the pinned pool per call
A = pkg.BinMatrix.random(65536, 65536)
for name, f in (("mzd_transpose(NULL, A) + mzd_free", lambda: L.mzd_free(L.mzd_transpose(None, A.mzd))),
                ("mzd_init + mzd_free alone", lambda: L.mzd_free(L.mzd_init(65536, 65536))),
                ("BinMatrix.transposed()", lambda: A.transposed())):
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    print("pipeline=%s %s: %.2f ms" % (os.environ.get("M4RI_HIP_TRANSPOSE_PIPELINE", "default"), name, best * 1e3), flush=True)
