#!/usr/bin/env python3
"""LPN-shaped tall-skinny products (BASELINE config 5): A (2^20 x 256) * X (256 x V), V in {1, 64, 128, 256}.

Two timings per V (development tool; `--json` writes one object per line for profiles/):
  warm: one A reused by every repetition -- 32 MiB, it stays in the 256 MiB Infinity Cache (MALL), so this is a cache figure;
  cold: NBUF (default 10) distinct A and C buffers visited round-robin -- 320 MiB of A between two uses of the same
        buffer, more than the MALL holds, so every repetition streams A from HBM.  This is the HBM-roofline figure.
Rates are algorithmic bytes (A once, X once, C once in the M4RI layout) per second against 8 TB/s."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import m4ri_rust_amd  # noqa
    from m4ri_rust_amd import _lib, device
    if os.environ.get("AB_LIB"):  # A/B of two builds on one box: load this shared object instead
        _lib.LIB_PATH = os.environ["AB_LIB"]
    device.require_gpu()
    as_json = "--json" in sys.argv
    nbuf = 10
    m, l = 1 << 20, 256
    cases = ((1, "naive"), (64, "naive"), (128, "naive"), (256, "naive"))
    if "--all" in sys.argv:
        cases += ((64, "m4rm"), (256, "m4rm"))
    stream = None
    if os.environ.get("LPN_STREAM"):  # time on a torch side stream (what bench.py does) instead of the NULL stream
        comp = torch.cuda.Stream()
        torch.cuda.set_stream(comp)
        stream = comp.cuda_stream
    if os.environ.get("LPN_PREALLOC_GIB"):  # an arena first, as in a process that has run the square products (placement of the buffers)
        hold = torch.empty(int(os.environ["LPN_PREALLOC_GIB"]) << 27, dtype=torch.int64, device="cuda")  # noqa: F841
    if os.environ.get("LPN_PREHEAT"):  # ten 65536^3 products first: the power / clock state a process is in after the square products
        P, Q = device.DMat.random(65536, 65536, 1), device.DMat.random(65536, 65536, 2)
        R = device.DMat(65536, 65536)
        for _ in range(int(os.environ["LPN_PREHEAT"])):
            device.mul(P, Q, C=R, stream=stream)
        torch.cuda.synchronize()
        del P, Q, R
    As = [device.DMat.random(m, l, 3 + i) for i in range(nbuf)]
    for V, algo in cases:
        X = device.DMat.random(l, V, 4)
        Cs = [device.DMat(m, V) for _ in range(nbuf)]
        res = {}
        for mode in ("warm", "cold"):
            k = 1 if mode == "warm" else nbuf
            for i in range(2 * k):
                device.mul(As[i % k], X, C=Cs[i % k], algo=algo, stream=stream)
            torch.cuda.synchronize()
            reps = 20 * nbuf
            t0 = time.perf_counter()
            for i in range(reps):
                device.mul(As[i % k], X, C=Cs[i % k], algo=algo, stream=stream)
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / reps
        wv = (V + 63) // 64
        alg = m * l / 8 + l * wv * 8 + m * wv * 8
        if as_json:
            print(json.dumps({"config": "5: LPN 2^20 x 256 times 256 x %d, %s entry" % (V, algo), "m": m, "l": l, "n": V,
                              "warm_us": res["warm"] * 1e6, "cold_us": res["cold"] * 1e6, "cold_buffers": nbuf,
                              "layout_bytes": alg, "cold_GBps": alg / res["cold"] / 1e9, "cold_frac_of_8TBps": alg / res["cold"] / 8e12,
                              "warm_GBps": alg / res["warm"] / 1e9}), flush=True)
        else:
            print(f"V={V:4d} algo={algo:6s} warm {res['warm']*1e6:7.1f} us ({alg/res['warm']/1e9:7.0f} GB/s)   "
                  f"cold {res['cold']*1e6:7.1f} us ({alg/res['cold']/1e9:7.0f} GB/s = {alg/res['cold']/8e12*100:.1f}% of 8 TB/s)", flush=True)
        del Cs


if __name__ == "__main__":
    main()
