#!/usr/bin/env python3
"""A grid of product shapes against a crude roofline -- max(operand bytes / 5 TB/s, bit-MACs / 7e15 per second of the 512-column tile
kernel, 6 us) -- to find shapes whose kernel is the wrong one (development tool).
    python tools/shape_holes.py [threshold=3] [b]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import m4ri_rust_amd  # noqa
from m4ri_rust_amd import device as dev
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
if len(sys.argv) > 2 and sys.argv[2] == "b":  # a second grid between the points of the first
    dims_m = [16, 128, 300, 2048, 9000, 33000, 200000]
    dims_l = [128, 600, 2048, 9000, 33000, 200000]
    dims_n = [16, 40, 100, 300, 600, 2048, 9000, 33000]
else:
    dims_m = [1, 8, 64, 1000, 4096, 20000, 65536]
    dims_l = [64, 256, 1000, 4096, 20000, 65536]
    dims_n = [1, 8, 64, 200, 256, 1000, 4096, 20000, 65536]
rows = []
for m in dims_m:
    for l in dims_l:
        for n in dims_n:
            if m * l * n > 65536 * 65536 * 4096 or (m * l + l * n + m * n) / 8 > 3e9:
                continue
            A, B, C = dev.DMat.random(m, l, 1), dev.DMat.random(l, n, 2), dev.DMat(m, n)
            for _ in range(3):
                dev.mul(A, B, C)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); dev.mul(A, B, C); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
            reps = max(3, min(500, int(0.02 / max(t1, 1e-6))))
            for _ in range(reps):
                dev.mul(A, B, C)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                dev.mul(A, B, C)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            wl, wn = (l + 63) // 64, (n + 63) // 64
            byt = 8.0 * (m * wl + l * wn + m * wn)
            roof = max(byt / 5e12, m * l * max(n, 1) / 7e15, 6e-6)
            rows.append((dt / roof, m, l, n, dt, roof))
            del A, B, C
rows.sort(reverse=True)
for r in rows:
    if r[0] >= thr:
        print("%6.1fx  %6d x %6d x %6d  %10.1f us  (roof %8.1f us)" % (r[0], r[1], r[2], r[3], r[4] * 1e6, r[5] * 1e6))
print("%d shapes, %d at %.1fx or more above the crude roofline" % (len(rows), sum(1 for r in rows if r[0] >= thr), thr))
