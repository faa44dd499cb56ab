#!/bin/bash
# One line per box: how the same binary measures on whichever MI355X this gpurun call landed on (the pool's boxes differ by up to 7 %,
# by more on LDS-bound microsecond kernels).  Appends to gpurun_out/r05_box_spread.txt.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
id=$(rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id" | head -1 | awk "{print \$NF}")
{
echo "box $(hostname) gpu ${id:-?} $(date -u +%H:%M:%S)"
python3 tools/shape_bench.py 65536,65536,65536,auto,0,5 32768,32768,32768,auto,0,10 4096,4096,4096,m4rm,0,200 2>/dev/null | grep -v amdgpu
CONFIGS_LPN_ONLY=1 CONFIGS_LPN_V=1,64,128,256 python3 tools/configs_only.py 2>/dev/null | grep "^config"
python3 tools/av_breakdown.py 2>/dev/null | grep "operator"
python3 tools/elim_bench.py 4096 65536 --cpu-max 0 2>/dev/null | grep "^n="
} >> gpurun_out/r05_box_spread.txt 2>&1
tail -14 gpurun_out/r05_box_spread.txt
