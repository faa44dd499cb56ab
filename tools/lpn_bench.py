#!/usr/bin/env python3
"""LPN-shaped tall-skinny products (BASELINE config 5): A (2^20 x 256) * X (256 x V), V in {1, 64, 256}.
Prints achieved algorithmic GB/s per product against the HBM roofline (development tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import m4ri_rust_amd  # noqa
from m4ri_rust_amd import device

m, l = 1 << 20, 256
A = device.DMat.random(m, l, 3)
for V, algo in ((1, "naive"), (64, "naive"), (64, "m4rm"), (128, "naive"), (256, "m4rm"), (256, "naive")):
    X = device.DMat.random(l, V, 4)
    C = device.DMat(m, V)
    for _ in range(3):
        device.mul(A, X, C=C, algo=algo)
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        device.mul(A, X, C=C, algo=algo)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    alg = m * l / 8 + l * ((V + 63) // 64) * 8 + m * ((V + 63) // 64) * 8
    print(f"V={V:4d} algo={algo:6s} {dt*1e6:9.1f} us  {alg/dt/1e9:8.1f} GB/s algorithmic ({alg/dt/8e12*100:.1f}% of 8 TB/s)  {2*m*l*V/dt/1e12:.2f} Tbit-ops/s")
