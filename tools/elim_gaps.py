#!/usr/bin/env python3
"""Launch-to-launch picture of the fused elimination steps from a rocprofv3 kernel trace (development tool):
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/eg -- python3 $REPO/tools/elim_bench.py 4096 --cpu-max 0 --reps 1
    python3 tools/elim_gaps.py /tmp/eg
Prints, for gf2_elim_update_kernel<true>: the launches' durations and the gaps between the end of one and the start of the next
when the next launch in the trace is again a fused step (inside a block)."""
import csv, glob, sys
paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for p in paths:
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur, gap = [], []
for a, b in zip(rows, rows[1:]):
    if "gf2_elim_update_kernel<true>" in a["Kernel_Name"]:
        dur.append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
        if "gf2_elim_update_kernel<true>" in b["Kernel_Name"]:
            gap.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
def q(v, f):
    v = sorted(v)
    return v[int(f * (len(v) - 1))] if v else 0
print("fused steps: %d; duration ns p10/p50/p90 = %d / %d / %d; gap to the next fused step ns p10/p50/p90 = %d / %d / %d (n = %d)" % (
    len(dur), q(dur, .1), q(dur, .5), q(dur, .9), q(gap, .1), q(gap, .5), q(gap, .9), len(gap)))
