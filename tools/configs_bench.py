"""Device-resident timings of the BASELINE.json configurations that fit one GPU (configs 2, 3, 5; config 4's
single-GPU form is bench.py itself).  Writes one JSON object per line; `python tools/configs_bench.py > profiles/...`"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import __graft_entry__ as ge
    ge.build()
    import torch
    import m4ri_rust_amd  # noqa: F401
    from m4ri_rust_amd import device as dev

    def run(name, m, l, n, algo, reps):
        A, B, C = dev.DMat.random(m, l, 1), dev.DMat.random(l, n, 2), dev.DMat(m, n)
        for _ in range(3):
            dev.mul(A, B, C, algo=algo)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            dev.mul(A, B, C, algo=algo)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        alg_bytes = (m * l + l * n + m * n) / 8
        # bytes the operands occupy in the M4RI layout (rows padded to whole 64-bit words): what a kernel must move at least
        layout_bytes = 8.0 * (m * ((l + 63) // 64) + l * ((n + 63) // 64) + m * ((n + 63) // 64))
        print(json.dumps({"config": name, "m": m, "l": l, "n": n, "algo": algo, "ms": dt * 1e3,
                          "bit_ops_per_s": 2.0 * m * l * n / dt, "algorithmic_GBps": alg_bytes / dt / 1e9,
                          "layout_GBps": layout_bytes / dt / 1e9, "layout_frac_of_8TBps": layout_bytes / dt / 8e12,
                          "strassen_levels": dev._lib.lib().gf2_strassen_levels(m, l, n, dev.ALGOS[algo], 0)}), flush=True)

    run("2: 4096^3, M4RM kernel only", 4096, 4096, 4096, "m4rm", 200)
    run("2b: 4096^3, automatic", 4096, 4096, 4096, "auto", 200)
    run("3: 32768^3, Strassen over M4RM", 32768, 32768, 32768, "auto", 10)
    run("3b: 32768^3, M4RM only", 32768, 32768, 32768, "m4rm", 5)
    for v in (1, 64, 128, 256):
        run("5: LPN 2^20 x 256 times 256 x %d, naive entry" % v, 1 << 20, 256, v, "naive", 200)


if __name__ == "__main__":
    main()
