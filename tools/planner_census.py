#!/usr/bin/env python3
"""Which tile-kernel variant the launch planner picks over a grid of shapes (runs without a GPU: plan_tiles is host code).
Evidence for the retirement rule: a kernel family that no shape of the grid selects leaves the shipped library."""
import ctypes, itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import m4ri_rust_amd  # noqa
from m4ri_rust_amd import _lib
L = _lib.lib()
out = (ctypes.c_longlong * 9)()
ms = [64, 128, 256, 300, 512, 1000, 2048, 3000, 4096, 6000, 8192, 12288, 16384, 20000, 32768, 65536]
ls = [64, 256, 512, 1024, 2048, 4096, 8192, 16384, 33000, 65536]
ns = [257, 300, 512, 600, 1024, 2048, 4096, 8192, 16384, 65536]
names = {7: "v3 (1024 x 2048)", 20: "v3, 4 waves (m <= 256)", 8: "v6 (2048 x 1024)", 9: "v8 4096 rows", 10: "v8 2048 rows", 11: "v8 1024 rows", 12: "v8 512 rows"}
count, examples = {}, {}
for m, l, n in itertools.product(ms, ls, ns):
    for batch, packed in ((1, 0), (1, 1), (49, 0), (49, 1)):
        if batch > 1 and (m > 8192 or l > 8192 or n > 8192):
            continue
        L.gf2_tile_plan(m, l, n, batch, packed, out)
        cfg = int(out[0])
        count[cfg] = count.get(cfg, 0) + 1
        examples.setdefault(cfg, []).append((m, l, n, batch, packed))
total = sum(count.values())
for cfg in sorted(count):
    ex = examples[cfg]
    print("%-26s %5d of %d plans   e.g. %s" % (names.get(cfg, "cfg %d" % cfg), count[cfg], total, ", ".join("%dx%dx%d%s%s" % (e[0], e[1], e[2], " x%d" % e[3] if e[3] > 1 else "", " packed" if e[4] else "") for e in ex[:: max(1, len(ex) // 4)][:4])))
