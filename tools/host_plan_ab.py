"""mzd_mul on host mzd_t under every schedule of the host path (M4RI_HIP_HOST_PLAN is read per call): 1 row blocks, 2 four equal
slabs of the inner dimension, 3 growing slabs, 4 two slabs, 0 the model's choice (development tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import m4ri_rust_amd as pkg
L = pkg._lib.lib()
for n in [int(a) for a in sys.argv[1:]] or [32768, 65536]:
    import ctypes
    te = (ctypes.c_double * 12)()
    os.environ["M4RI_HIP_HOST_PLAN"] = "0"
    ch = L.gf2_host_plan_model(n, n, n, 0, 0, te)
    print("n=%d model: %s ms -> schedule %d" % (n, ", ".join("%d: %.2f" % (i + 1, te[i] * 1e3) for i in range(12) if te[i] > 0), ch), flush=True)
    A, B = pkg.BinMatrix.random(n, n), pkg.BinMatrix.random(n, n)
    C = pkg.BinMatrix.zero(n, n)
    for plan in (1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 0, 1, 0):
        os.environ["M4RI_HIP_HOST_PLAN"] = str(plan)
        L.mzd_mul(C.mzd, A.mzd, B.mzd, 0)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); L.mzd_mul(C.mzd, A.mzd, B.mzd, 0); ts.append(time.perf_counter() - t0)
        print("n=%d plan %d: min %.2f ms  median %.2f ms" % (n, plan, min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3), flush=True)
    del A, B, C
