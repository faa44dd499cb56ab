"""Where the time of `&A * &v` (2^20 x 256) goes through the C ABI, call by call (development tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import m4ri_rust_amd as pkg
L = pkg._lib.lib()
m, l = 1 << 20, 256
A = pkg.BinMatrix.random(m, l)
vrow = pkg.BinMatrix.random(1, l)
def once(acc):
    t = [time.perf_counter()]
    vt = L.mzd_transpose(None, vrow.mzd); t.append(time.perf_counter())
    r = L.mzd_mul_naive(None, A.mzd, vt); t.append(time.perf_counter())
    rt = L.mzd_transpose(None, r); t.append(time.perf_counter())
    L.mzd_free(vt); L.mzd_free(r); L.mzd_free(rt); t.append(time.perf_counter())
    r2 = L.mzd_init(m, 1); t.append(time.perf_counter())
    L.mzd_free(r2); t.append(time.perf_counter())
    for i in range(len(t) - 1): acc[i] = min(acc[i], t[i + 1] - t[i])
names = ["mzd_transpose(v)", "mzd_mul_naive(NULL, A, vT)", "mzd_transpose(result 2^20 x 1)", "3 x mzd_free", "mzd_init(2^20, 1)", "mzd_free"]
for cached in (True, False):
    if cached: L.gf2_mzd_cache_on_device(A.mzd)
    else: L.gf2_mzd_uncache(A.mzd)
    acc = [1e9] * 6
    for _ in range(8): once(acc)
    print("A %s" % ("cached on the device (gf2_mzd_cache_on_device)" if cached else "uploaded by every call"))
    for n, a in zip(names, acc): print("  %-40s %8.1f us" % (n, a * 1e6))
    print("  %-40s %8.1f us" % ("the operator (first four lines)", sum(acc[:4]) * 1e6))
