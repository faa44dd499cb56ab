#!/usr/bin/env python3
"""Which clock does the chip hold under the M4RM tile kernel?  (development tool, needs tools/libm4ri_hip_dev.so)

    AB_LIB=tools/libm4ri_hip_dev.so python3 tools/clock_probe.py [n] [> profiles/rNN_clock.jsonl]

For each bit density of profiles/r04_density.jsonl (half / sparse / ones): >= 2 s of back-to-back n^3 products on resident operands, then
the stamps of the LAST product's tile launches -- every workgroup of gf2_m4rm_kernel_v8 (development build) stores s_memrealtime /
s_memtime at its start and end; clock = d(s_memtime) / d(s_memrealtime) x 100 MHz inside one workgroup, median over workgroups
(MI355X_MICROARCH.md, DVFS give-back item 6).  Prints one JSON line per density and, last, the summary bench.py replays
(profiles/clock_latest.json is written when argv has --write)."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import m4ri_rust_amd as pkg
from m4ri_rust_amd import _lib, device
if os.environ.get("AB_LIB"):
    _lib.LIB_PATH = os.environ["AB_LIB"]
L = _lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
if not hasattr(raw, "gf2k_dev_set_clock_stamps"):
    sys.exit("clock_probe: needs the development library (AB_LIB=tools/libm4ri_hip_dev.so)")
raw.gf2k_dev_set_clock_stamps.argtypes = [ctypes.c_void_p]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 65536
ldw = n // 64
A_t = torch.empty((n, ldw), dtype=torch.int64, device="cuda")
B_t = torch.empty((n, ldw), dtype=torch.int64, device="cuda")
C_t = torch.empty((n, ldw), dtype=torch.int64, device="cuda")
A, B, C = (device.DMat.from_torch(t, n) for t in (A_t, B_t, C_t))
stamps = torch.zeros(2 << 20, dtype=torch.int64, device="cuda")
out = []
for density in ("half", "sparse", "ones"):
    A.fill_random(1), B.fill_random(2)
    if density == "ones":
        A_t.fill_(-1), B_t.fill_(-1)
    elif density == "sparse":  # AND of six independent fills: density 2^-6 (as bench.py --density sparse)
        tmp_t = torch.empty_like(A_t)
        tmp = device.DMat.from_torch(tmp_t, n)
        for t_, base in ((A_t, 100), (B_t, 200)):
            for k in range(5):
                tmp.fill_random(base + k)
                t_.bitwise_and_(tmp_t)
        del tmp, tmp_t
    torch.cuda.synchronize()
    assert raw.gf2k_dev_set_clock_stamps(None) == 0
    device.mul(A, B, C=C, algo="auto")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 2.5:  # clocks settle under the load
        device.mul(A, B, C=C, algo="auto")
        reps += 1
        if reps % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt_plain = []
    for _ in range(5):
        t1 = time.perf_counter()
        device.mul(A, B, C=C, algo="auto")
        torch.cuda.synchronize()
        dt_plain.append(time.perf_counter() - t1)
    stamps.zero_()
    torch.cuda.synchronize()
    assert raw.gf2k_dev_set_clock_stamps(stamps.data_ptr()) == 0
    t1 = time.perf_counter()
    device.mul(A, B, C=C, algo="auto")
    torch.cuda.synchronize()
    dt_stamped = time.perf_counter() - t1
    assert raw.gf2k_dev_set_clock_stamps(None) == 0
    h = stamps.cpu().numpy().view(np.uint64).reshape(-1, 4)
    res = {"density": density, "n": n, "ms_per_product": min(dt_plain) * 1e3, "ms_stamped_product": dt_stamped * 1e3}
    for name, lo, hi in (("tall_packed_v8<8,*,1>", 0, 1 << 18), ("other_v8", 1 << 18, 2 << 18)):
        g = h[lo:hi]
        g = g[(g[:, 0] != 0) & (g[:, 3] > g[:, 0])]
        if not len(g):
            continue
        dc = (g[:, 2] - g[:, 1]).astype(np.float64)
        drt = (g[:, 3] - g[:, 0]).astype(np.float64)
        clk = dc / (drt * 10.0)  # cycles per ns
        span = (g[:, 3].max() - g[:, 0].min()) / 100.0  # us, all workgroups of the last launch of this kind
        res[name] = {"workgroups": int(len(g)), "clock_GHz_median": float(np.median(clk)), "clock_GHz_p10": float(np.percentile(clk, 10)),
                     "clock_GHz_p90": float(np.percentile(clk, 90)), "workgroup_us_median": float(np.median(drt) / 100.0),
                     "workgroup_cycles_median": float(np.median(dc)), "launch_span_us": float(span)}
    out.append(res)
    print(json.dumps(res), flush=True)
half = out[0].get("tall_packed_v8<8,*,1>", {})
summary = {"n": n, "kernel": "gf2_m4rm_kernel_v8<8,2,1,2,1>", "clock_GHz": {r["density"]: r.get("tall_packed_v8<8,*,1>", {}).get("clock_GHz_median") for r in out},
           "ms_per_product": {r["density"]: r["ms_per_product"] for r in out},
           "method": "d(s_memtime)/d(s_memrealtime) x 100 MHz per workgroup, median over the workgroups of the last tile launch after >= 2.5 s of "
                     "back-to-back products (tools/clock_probe.py, development build of the kernel; the shipped kernel executes no stamp)"}
print(json.dumps(summary))
if "--write" in sys.argv:
    with open(os.path.join(ROOT, "profiles", "clock_latest.json"), "w") as f:
        json.dump(summary, f, indent=1)
