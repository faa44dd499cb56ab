#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R/tools
{
echo "== 343 leaves of 4096^3, packed A"; APACK=1 ./kbench 4096 343 3 90 9 10 11 12
echo "== 343 leaves of 4096^3, unpacked A"; ./kbench 4096 343 3 9 10 11 12
echo "== 4096^3 single, all tiles split (unpacked)"; NREM=-1 ./kbench 4096 1 200 9 10 11 12
echo "== 4096^3 single, all tiles split (packed, pack not timed)"; APACK=1 NREM=-1 ./kbench 4096 1 200 9 10 11 12
echo "== 2048^3 single"; NREM=-1 ./kbench 2048 1 200 10 11 12
echo "== 8192^3 single"; NREM=-1 ./kbench 8192 1 50 9 10 11 12; APACK=1 NREM=-1 ./kbench 8192 1 50 9 10 11
} > $O/calib2.txt 2>&1
echo calib2 done
