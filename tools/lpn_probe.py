#!/usr/bin/env python3
"""Why does bench.py's LPN leg time V = 256 slower than tools/lpn_bench.py?  Same process, same buffers, both loop styles (development tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import m4ri_rust_amd  # noqa
from m4ri_rust_amd import device
m, l, nbuf = 1 << 20, 256, 10
use_stream = bool(os.environ.get("PROBE_STREAM"))
stream = None
if use_stream:
    comp = torch.cuda.Stream(); torch.cuda.set_stream(comp); stream = comp.cuda_stream
seeds = [1] + [100 + i for i in range(1, nbuf)] if os.environ.get("PROBE_BENCH_SEEDS") else [3 + i for i in range(nbuf)]
As = [device.DMat.random(m, l, sd, stream) for sd in seeds]
for V in (64, 256):
    X = device.DMat.random(l, V, 2, stream)
    Cs = [device.DMat(m, V) for _ in range(nbuf)]
    for rnd in range(3):
        for i in range(100):
            device.mul(As[i % nbuf], X, C=Cs[i % nbuf], algo="naive", stream=stream)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(200):
            device.mul(As[i % nbuf], X, C=Cs[i % nbuf], algo="naive", stream=stream)
        torch.cuda.synchronize(); a = (time.perf_counter() - t0) / 200
        fn = lambda i: device.mul(As[i % nbuf], X, C=Cs[i % nbuf], algo="naive", stream=stream)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(200):
            fn(i)
        torch.cuda.synchronize(); b = (time.perf_counter() - t0) / 200
        # events around the loop (GPU time only)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200):
            device.mul(As[i % nbuf], X, C=Cs[i % nbuf], algo="naive", stream=stream)
        e1.record(); torch.cuda.synchronize()
        print("V=%d round %d: loop %.2f us, lambda %.2f us, events %.2f us; C ptrs mod 2MiB: %s" % (
            V, rnd, a * 1e6, b * 1e6, e0.elapsed_time(e1) * 1e3 / 200,
            [hex(int(c.s.data or 0) % (1 << 30)) for c in Cs[:3]] + [hex(int(a_.s.data or 0) % (1 << 30)) for a_ in As[:3]]), flush=True)
    del Cs, X
