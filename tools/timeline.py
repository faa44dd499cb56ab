#!/usr/bin/env python3
"""Prints the last `span_ms` of a rocprofv3 --kernel-trace --memory-copy-trace run as one merged timeline (development tool)."""
import csv, glob, sys
d = sys.argv[1]
span = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
min_ns = float(sys.argv[3]) if len(sys.argv) > 3 else 20000.0  # shortest event shown
ev = []
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:60]))
for f in glob.glob(d + "/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s" % (r.get("Direction", ""), r.get("Size", r.get("Bytes", "")))))
ev.sort()
end = max(e[1] for e in ev)
t0 = end - span * 1e6
for s, e, nm in ev:
    if e >= t0 and (e - s) > min_ns:
        print("%8.3f -> %8.3f ms (%7.3f)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, nm))
