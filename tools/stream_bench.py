#!/usr/bin/env python3
"""Cold streaming rate of short kernels (development tool): C = A ^ B on m x 256-bit matrices (32 bytes per row), with the operands
rotated through NBUF distinct buffers so that nothing is served from the Infinity Cache.  Reference point for the LPN kernels:
what a plain streaming kernel of the same size reaches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import m4ri_rust_amd  # noqa
from m4ri_rust_amd import device
device.require_gpu()
nbuf = 12
for logm in (18, 19, 20, 21, 22, 24):
    m = 1 << logm
    As = [device.DMat.random(m, 256, 3 + i) for i in range(nbuf)]
    Cs = [device.DMat(m, 256) for _ in range(nbuf)]
    for mode, k in (("warm", 1), ("cold", nbuf)):
        for i in range(2 * k):
            device.add(As[i % k], As[(i + 1) % k], Cs[i % k])
        torch.cuda.synchronize()
        reps = 10 * nbuf
        t0 = time.perf_counter()
        for i in range(reps):
            device.add(As[i % k], As[(i + 1) % k], Cs[i % k])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        b = 3 * m * 32
        print(f"m=2^{logm} ({m*32/2**20:.0f} MiB per operand) {mode}: {dt*1e6:8.1f} us  {b/dt/1e9:7.0f} GB/s", flush=True)
    del As, Cs
