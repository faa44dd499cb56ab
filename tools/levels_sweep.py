#!/usr/bin/env python3
"""Times every Strassen level count for a few shapes and shows what the automatic choice picks (development tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import m4ri_rust_amd  # noqa
from m4ri_rust_amd import device, sharded

shapes = [(8192, 8192, 8192), (16384, 16384, 16384), (32768, 32768, 32768), (8192, 65536, 16384), (16384, 65536, 16384),
          (32768, 65536, 16384), (8192, 65536, 65536), (65536, 65536, 65536)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split('x')) for a in sys.argv[1:]]
for (m, l, n) in shapes:
    A, B = device.DMat.random(m, l, 1), device.DMat.random(l, n, 2)
    C = device.DMat(m, n)
    res = []
    for L in range(0, 6):
        algo = "m4rm" if L == 0 else "strassen"
        if L and sharded.levels_used(m, l, n, "strassen", L) != L:
            break
        for _ in range(2):
            device.mul(A, B, C=C, algo=algo, param=L)
        torch.cuda.synchronize()
        reps = 3 if m * l * n > 1e13 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            device.mul(A, B, C=C, algo=algo, param=L)
        torch.cuda.synchronize()
        res.append((L, (time.perf_counter() - t0) / reps * 1e3, device._lib.lib().gf2_model_time(m, l, n, L) * 1e3))
    auto = sharded.levels_used(m, l, n, "auto", 0)
    best = min(res, key=lambda x: x[1])
    print(f"{m}x{l}x{n}: " + "  ".join(f"L{L}={t:.3f}ms (model {mt:.3f})" for L, t, mt in res) + f"   auto={auto} best={best[0]}", flush=True)
    del A, B, C
