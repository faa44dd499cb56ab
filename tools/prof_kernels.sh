#!/bin/bash
# usage (on the GPU box): tools/prof_kernels.sh <tag> <command...>   -> per-kernel average durations of the command
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- "$@" > $R/gpurun_out/prof_$tag.log 2>&1
cd $R
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/prof_%s/*/*kernel_stats.csv' % sys.argv[1])[0]
print("== %s" % sys.argv[1])
for r in csv.DictReader(open(f)):
    print("%-70s calls %4s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
