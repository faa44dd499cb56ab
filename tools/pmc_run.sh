#!/bin/bash
# usage (on the GPU box): tools/pmc_run.sh <tag> "<counters (<= 4)>" <command...>  -> per-kernel average of each counter
# (counter collection only: --kernel-trace, no other trace domain, as the pool requires)
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
cd $R
python3 - "$tag" <<'PY'
import csv, glob, sys
from collections import defaultdict
f = glob.glob('gpurun_out/pmc_%s/*/*counter_collection.csv' % sys.argv[1])[0]
per = defaultdict(float); kern = {}
for r in csv.DictReader(open(f)):
    key = (r["Dispatch_Id"], r["Counter_Name"]); per[key] += float(r["Counter_Value"]); kern[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
agg = defaultdict(list)
for (d, c), v in per.items(): agg[(kern[d], c)].append(v)
print("== %s" % sys.argv[1])
for (k, c), v in sorted(agg.items()):
    print("%-60s %-24s n=%4d avg %16.1f" % (k[:60], c, len(v), sum(v) / len(v)))
PY
