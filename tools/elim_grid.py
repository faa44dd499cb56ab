#!/usr/bin/env python3
"""Echelon forms over a grid of shapes, to spot anomalies (development tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import m4ri_rust_amd  # noqa
from m4ri_rust_amd import device as dev
for m in (64, 1000, 4096, 20000, 65536, 262144):
    for n in (64, 1000, 4096, 20000, 65536):
        if m * n > 65536 * 65536:
            continue
        A = dev.DMat.random(m, n, 1)
        ts = []
        for full in (0, 1):
            B = A.clone(); torch.cuda.synchronize()
            t0 = time.perf_counter(); r = dev.echelonize(B, full=bool(full))[0]; torch.cuda.synchronize()
            B = A.clone(); torch.cuda.synchronize()
            t0 = time.perf_counter(); r = dev.echelonize(B, full=bool(full))[0]; torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print("%7d x %6d rank %6d  upper %9.3f ms  reduced %9.3f ms" % (m, n, r, ts[0] * 1e3, ts[1] * 1e3), flush=True)
