#!/bin/bash
# Collects round 4's measurement evidence on the GPU box into gpurun_out/r04f/ (copied into profiles/r04_* afterwards):
#   tools/collect_profiles_r04.sh        (run from the repo root through gpurun)
# rocprofv3 runs from /tmp with TMPDIR=/tmp, the program directly after `--`; counter passes use --kernel-trace only.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err
echo "bench done"
python3 tools/lpn_bench.py --json 2>/dev/null | grep "^{" > $O/lpn_cold_warm.jsonl
CONFIGS_LPN_ONLY=1 CONFIGS_LPN_V=1,64,128,256 python3 tools/configs_only.py 2>/dev/null | grep "^config" > $O/lpn_bench_leg.txt
tools/lpn_lab 64 128 256 > $O/lpn_lab.txt 2>&1
python3 tools/host_path_bench.py 2>/dev/null | grep "^n=" > $O/host_path.txt
python3 tools/pcie_duplex.py 2>/dev/null | grep -v amdgpu >> $O/host_path.txt
python3 tools/transpose_bench.py 2>/dev/null | grep -v amdgpu > $O/transpose.txt
echo "lpn / host path / transpose done"
python3 tools/levels_sweep.py 4096x4096x4096 8192x8192x8192 10240x10240x10240 12288x12288x12288 14336x14336x14336 16384x16384x16384 20480x20480x20480 \
   24576x24576x24576 28672x28672x28672 32768x32768x32768 40960x40960x40960 49152x49152x49152 65536x65536x65536 8192x65536x65536 16384x65536x65536 \
   16384x65536x16384 8192x65536x16384 8192x65536x32768 2>/dev/null | grep -v amdgpu > $O/levels_sweep.txt
echo "levels sweep done"
python3 tools/shape_bench.py 8192,65536,65536,auto,0,10 8192,65536,32768,auto,0,10 8192,65536,16384,auto,0,20 16384,65536,65536,auto,0,5 16384,65536,32768,auto,0,10 \
   16384,65536,16384,auto,0,10 32768,65536,65536,auto,0,5 32768,65536,32768,auto,0,5 32768,65536,16384,auto,0,10 65536,65536,65536,auto,0,3 65536,65536,65536,m4rm,0,3 \
   60000,60000,60000,auto,0,3 70000,70000,70000,auto,0,3 32768,32768,32768,auto,0,10 16384,16384,16384,auto,0,20 4096,4096,4096,m4rm,0,200 2>/dev/null | grep -v amdgpu > $O/size_sweep.txt
echo "size sweep done"
python3 tools/elim_bench.py 2>/dev/null | grep "^n=" > $O/elim.txt
echo "elimination done"
python3 tools/host_transpose_bench.py 2>/dev/null | grep "^pipeline" >> $O/host_path.txt
tools/tilecopy 65536 > $O/tilecopy.txt 2>&1
echo "timings done"
# density sanity runs (sparse 1/64 and all ones, clock / DVFS)
for d in sparse ones half; do python3 bench.py --steps 10 --warmup 3 --no-cpu --no-configs --no-host-path --no-elim --density $d 2>/dev/null | grep "^{" >> $O/density.jsonl; done
echo "density done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-configs --no-host-path --no-elim > $O/prof_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-configs --no-parity --no-host-path --no-elim > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-configs --no-parity --no-host-path --no-elim > $O/pmc_write.log 2>&1
echo "bench profiles done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lpn -- python3 $R/tools/lpn_pmc.py > $O/prof_lpn.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_lpn_fetch -- python3 $R/tools/lpn_pmc.py > $O/pmc_lpn_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_lpn_write -- python3 $R/tools/lpn_pmc.py > $O/pmc_lpn_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_configs -- python3 $R/tools/configs_only.py > $O/prof_configs.log 2>&1
echo "config profiles done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_elim -- python3 $R/tools/elim_bench.py 4096 65536 --cpu-max 0 > $O/prof_elim.log 2>&1
echo "elimination profile done"
cd $R
cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/final_kernel_stats.csv
cp $(ls $O/prof_lpn/*/*kernel_stats.csv | head -1) $O/lpn_kernel_stats.csv
cp $(ls $O/prof_configs/*/*kernel_stats.csv | head -1) $O/configs_kernel_stats.csv
cp $(ls $O/prof_elim/*/*kernel_stats.csv | head -1) $O/elim_kernel_stats.csv
python3 tools/pmc_summary.py FETCH_SIZE=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) WRITE_SIZE=$(ls $O/pmc_write/*/*counter_collection.csv | head -1) > $O/final_pmc_summary.json
python3 tools/pmc_summary.py FETCH_SIZE=$(ls $O/pmc_lpn_fetch/*/*counter_collection.csv | head -1) WRITE_SIZE=$(ls $O/pmc_lpn_write/*/*counter_collection.csv | head -1) > $O/lpn_pmc_summary.json
rm -rf $O/prof_elim $O/prof_bench $O/pmc_fetch $O/pmc_write $O/prof_lpn $O/pmc_lpn_fetch $O/pmc_lpn_write $O/prof_configs
# SQ counters of the two LPN kernels (lab variants = the shipped configurations)
tools/lab_pmc.sh 256 "lpn256<512,8,m2,e3,o13>" > $O/lpn256_sq.txt 2>&1
tools/lab_pmc.sh 64 "lpn8<w1,512,4,m2,e3,strided>" > $O/lpn8_sq.txt 2>&1
python3 bench.py --gpus 2 --backend gloo --check --dim 16384 --no-cpu --steps 3 --warmup 1 2>$O/two_rank_rehearsal.err | grep "^{" > $O/two_rank_rehearsal.json
ls -la $O
