#!/usr/bin/env python3
"""Per-wave wall-clock stamps of gf2_tallskinny5_kernel (development build, tools/libm4ri_hip_dev.so): where the time of a
cold 2^20 x 256 times 256 x V product goes.  python tools/ts_stamps.py [V]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
V = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libm4ri_hip_dev.so"))
m, l = 1 << 20, 256
nbuf = 10
As = [torch.randint(-2**62, 2**62, (m, 4), dtype=torch.int64, device="cuda") for _ in range(nbuf)]
W = (V + 63) // 64
B = torch.randint(-2**62, 2**62, (256, W), dtype=torch.int64, device="cuda")
Cs = [torch.empty((m, W), dtype=torch.int64, device="cuda") for _ in range(nbuf)]
waves = (m // 1024) * 16
st = torch.zeros((waves, 8), dtype=torch.int64, device="cuda")
f = lib.gf2k_tallskinny5_dbg
f.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong,
              ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for it in range(3 * nbuf):
    i = it % nbuf
    assert f(As[i].data_ptr(), 4, B.data_ptr(), W, Cs[i].data_ptr(), W, m, l, V, st.data_ptr(), None) == 0
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.int64)
s = s[s[:, 0] != 0]
t0 = s[:, 0].min()
us = (s - t0) / 100.0
names = ["start", "built(ph0)", "row0 done(ph0)", "last row done(ph0)", "built(ph1)", "row0 done(ph1)", "last row done(ph1)"]
import time
torch.cuda.synchronize(); t1 = time.perf_counter()
for it in range(10 * nbuf):
    i = it % nbuf
    f(As[i].data_ptr(), 4, B.data_ptr(), W, Cs[i].data_ptr(), W, m, l, V, st.data_ptr(), None)
torch.cuda.synchronize()
print("V=%d variant %s: cold %.1f us per call" % (V, os.environ.get("TS5_VARIANT", "default"), (time.perf_counter() - t1) / (10 * nbuf) * 1e6))
for k, nm in enumerate(names):
    col = us[:, k]
    if (s[:, k] == 0).all():
        continue
    print("%-20s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (nm, col.min(), np.median(col), np.percentile(col, 90), col.max()))
