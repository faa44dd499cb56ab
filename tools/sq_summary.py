"""SQ counters of the tile kernel from several rocprofv3 --pmc passes (csv): python tools/sq_summary.py <dir> <dir> ... > json
Sums each counter over the instances of a dispatch, averages over the dispatches of the kernel whose name contains 'gf2_m4rm_kernel_v8'."""
import csv
import glob
import json
import sys
from collections import defaultdict

per = defaultdict(float)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        vals = defaultdict(float)
        for row in csv.DictReader(open(f, newline="")):
            if "gf2_m4rm_kernel_v8" in row["Kernel_Name"]:
                vals[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        byc = defaultdict(list)
        for (disp, c), v in vals.items():
            byc[c].append(v)
        for c, v in byc.items():
            per[c] = sum(v) / len(v)
quads = 32 * 2 * 16 * 256  # kbench 8192 32: 32 products x (2 x 16) tiles of 4096 x 512 x 256 quads (32 bits of the inner dimension)
out = {"source": "rocprofv3 --pmc <4 counters per pass> --kernel-trace --output-format csv -- tools/kbench 8192 32 1 9 with APACK=1 "
                 "(gf2_m4rm_kernel_v8<8,2,1,2,1>: 4096-row tiles on row-group-packed A, 32 products 8192^3 = 1024 workgroups x 256 quads)",
       "per_launch": dict(sorted(per.items()))}
if per.get("SQ_LDS_IDX_ACTIVE") and per.get("SQ_WAVE_CYCLES"):
    cu_quads = quads  # one workgroup (= one CU's worth of work) per tile
    out["derived"] = {
        "quads": quads,
        "lds_array_cycles_per_quad_per_cu": per["SQ_LDS_IDX_ACTIVE"] / cu_quads,
        "wave_cycles_per_quad": per["SQ_WAVE_CYCLES"] * 4 / (cu_quads * 8),
        "lds_pipe_busy": (per["SQ_LDS_IDX_ACTIVE"] / cu_quads) / (per["SQ_WAVE_CYCLES"] * 4 / (cu_quads * 8)),
        "lds_bank_conflict_share_of_lds_active": per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_LDS_IDX_ACTIVE"],
        "wait_any_share_of_wave_cycles": per.get("SQ_WAIT_ANY", 0.0) / per["SQ_WAVE_CYCLES"],
        "instructions_per_wave_per_quad": {k: per.get(c, 0.0) / (cu_quads * 8) for k, c in
                                           (("valu", "SQ_INSTS_VALU"), ("lds", "SQ_INSTS_LDS"), ("vmem_rd", "SQ_INSTS_VMEM_RD"), ("salu", "SQ_INSTS_SALU"))},
    }
json.dump(out, sys.stdout, indent=1)
print()
