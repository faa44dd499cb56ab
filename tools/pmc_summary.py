"""Per-kernel averages of rocprofv3 counter runs (csv output):
     python tools/pmc_summary.py FETCH_SIZE=<..._counter_collection.csv> WRITE_SIZE=<..._counter_collection.csv> > summary.json
   Values of one dispatch (one row per counter instance) are summed, then averaged over the kernel's dispatches."""
import csv
import json
import sys
from collections import defaultdict

out = {}
for arg in sys.argv[1:]:
    name, path = arg.split("=", 1)
    per_dispatch = defaultdict(float)
    kern = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != name:
                continue
            d = row["Dispatch_Id"]
            per_dispatch[d] += float(row["Counter_Value"])
            kern[d] = row["Kernel_Name"].split("(")[0]
    agg = defaultdict(list)
    for d, v in per_dispatch.items():
        agg[kern[d]].append(v)
    out[name] = {k: {"dispatches": len(v), "avg_KB_per_dispatch": sum(v) / len(v)} for k, v in agg.items()}
json.dump(out, sys.stdout, indent=1)
print()
