#!/usr/bin/env python3
"""How much does the placement of C relative to A matter for the tall-skinny kernels?  One 3 GiB allocation, A_i at
i * 64 MiB, C_i at 1.5 GiB + i * 64 MiB + delta; the cold rotation of tools/lpn_bench.py for each delta (development tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import m4ri_rust_amd  # noqa
from m4ri_rust_amd import device
m, l, nbuf = 1 << 20, 256, 10
MiB = 1 << 20
big = torch.empty(3 * 1024 * MiB // 8, dtype=torch.int64, device="cuda")
base = big.data_ptr()
print("base %x" % base)
As = [device.DMat.wrap(base + i * 64 * MiB, m, l, l // 64, keep=big) for i in range(nbuf)]
for i, a in enumerate(As):
    a.fill_random(3 + i)
for V in [int(v) for v in (sys.argv[1:] or ["256", "64"])]:
    X = device.DMat.random(l, V, 4)
    wv = (V + 63) // 64
    deltas = [0, 4096, 65536, MiB // 2, MiB] + [k * 2 * MiB for k in range(1, 17)]
    for d in deltas:
        Cs = [device.DMat.wrap(base + 1536 * MiB + i * 64 * MiB + d, m, V, wv, keep=big) for i in range(nbuf)]
        for i in range(50):
            device.mul(As[i % nbuf], X, C=Cs[i % nbuf], algo="naive")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(200):
            device.mul(As[i % nbuf], X, C=Cs[i % nbuf], algo="naive")
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        print("V=%d delta %8.3f MiB: %.2f us" % (V, d / MiB, dt * 1e6), flush=True)
