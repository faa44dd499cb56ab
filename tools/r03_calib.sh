#!/bin/bash
# round-3 calibration of the v8 tile-kernel family on the GPU box: per-variant times on batched leaves, small products with
# stream-K splits.  Output: gpurun_out/r03/calib.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R/tools
{
echo "== 343 leaves of 4096^3, packed A"; APACK=1 ./kbench 4096 343 3 90 9 10 11 12
echo "== 343 leaves of 4096^3, unpacked A"; ./kbench 4096 343 3 90 9 10 11 12 8
echo "== 343 leaves, packed, stream-K on the last round"; APACK=1 NREM=-2 ./kbench 4096 343 3 9 10
echo "== 49 leaves of 4096^3 packed (392 / 784 tiles), whole tiles vs stream-K"; APACK=1 ./kbench 4096 49 5 9 10; APACK=1 NREM=-2 ./kbench 4096 49 5 9 10
echo "== 4096^3 single, all tiles split (unpacked)"; NREM=-1 ./kbench 4096 1 200 9 10 11 12
echo "== 4096^3 single, all tiles split (packed, pack not timed)"; APACK=1 NREM=-1 ./kbench 4096 1 200 9 10 11 12
echo "== 4096^3 single, 512 segments"; NREM=-1 NSEG=512 ./kbench 4096 1 200 10 11 12
echo "== 4096^3 old kernels"; KSPLIT=32 ./kbench 4096 1 200 8 90; KSPLIT=8 ./kbench 4096 1 200 7
echo "== 2048^3 single"; NREM=-1 ./kbench 2048 1 200 10 11 12; KSPLIT=16 ./kbench 2048 1 200 7 8
echo "== 8192^3 single"; NREM=-1 ./kbench 8192 1 50 9 10 11 12; APACK=1 NREM=-1 ./kbench 8192 1 50 9 10 11
echo "== 16384^3 single"; NREM=-1 ./kbench 16384 1 10 9 10; APACK=1 NREM=-1 ./kbench 16384 1 10 9 10; APACK=1 ./kbench 16384 1 10 9 10
} > $O/calib.txt 2>&1
echo calib done
