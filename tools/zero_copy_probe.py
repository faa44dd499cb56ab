"""Can the LPN kernels read A from / write C into PINNED HOST memory directly (no separate copy)?  2^20 x 256 x V on the device API
with the data pointers of pinned mzd_t blocks; development probe."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import m4ri_rust_amd as pkg
from m4ri_rust_amd import device
L = pkg._lib.lib()
m, l = 1 << 20, 256
for V in (1, 64):
    A = pkg.BinMatrix.random(m, l)
    X = pkg.BinMatrix.random(l, V)
    C = pkg.BinMatrix.zero(m, V)
    za, zc = A.mzd.contents, C.mzd.contents
    pa = ctypes.cast(za.rows[0], ctypes.c_void_p).value
    pc = ctypes.cast(zc.rows[0], ctypes.c_void_p).value
    dA, dX = device.DMat.from_host(A), device.DMat.from_host(X)
    dC = device.DMat(m, V)
    hA = device.DMat.wrap(pa, m, l, za.rowstride)
    hC = device.DMat.wrap(pc, m, V, zc.rowstride)
    ref = device.mul(dA, dX, algo="naive").to_words()
    for an, a in (("A dev ", dA), ("A HOST", hA)):
        for cn, c in (("C dev ", dC), ("C HOST", hC)):
            ts = []
            for _ in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pkg._lib.check(L.gf2_mul_dev(ctypes.byref(c.s), ctypes.byref(a.s), ctypes.byref(dX.s), 0, 3, 0, None), "mul")
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            got = C.to_words() if c is hC else dC.to_words()
            print("V=%d %s %s: min %.1f us  ok=%s" % (V, an, cn, min(ts) * 1e6, np.array_equal(got, ref)), flush=True)
