#!/bin/bash
# Collects round 5's measurement evidence on the GPU box into gpurun_out/r05f/ (copied into profiles/r05_* afterwards):
#   tools/collect_profiles_r05.sh        (run from the repo root through gpurun; ~8 minutes)
# rocprofv3 runs from /tmp with TMPDIR=/tmp, the program directly after `--`; counter passes use --kernel-trace only.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err
echo "bench done"
python3 tools/av_breakdown.py 2>/dev/null | grep -v amdgpu > $O/av_breakdown.txt
python3 tools/host_plan_ab.py 32768 65536 16384 2>/dev/null | grep -v amdgpu > $O/host_plan_ab.txt
tools/lpn_lab 64 128 256 > $O/lpn_lab_final.txt 2>&1
python3 tools/elim_bench.py 2>/dev/null | grep "^n=" > $O/elim.txt
AB_LIB=tools/libm4ri_hip_dev.so python3 tools/elim_stamps.py 4096 65536 2>/dev/null | grep -v amdgpu > $O/elim_stamps.txt
echo "host path / lab / elimination done"
# the per-rank shapes of the multi-GPU step on R x Q grids (DESIGN section 6's cap table): N = 8: 8x1, 4x2, 2x4, 1x8 and their halves
# (two sub-panels); N = 4: 4x1, 2x2, 1x4; N = 2: 2x1, 1x2; the whole product
python3 tools/shape_bench.py 65536,65536,65536,auto,0,3 8192,65536,65536,auto,0,10 16384,65536,32768,auto,0,10 32768,65536,16384,auto,0,10 65536,65536,8192,auto,0,10 \
   16384,65536,16384,auto,0,10 32768,65536,8192,auto,0,10 65536,65536,4096,auto,0,10 8192,65536,32768,auto,0,10 \
   16384,65536,65536,auto,0,5 32768,65536,32768,auto,0,5 65536,65536,16384,auto,0,5 32768,65536,65536,auto,0,5 65536,65536,32768,auto,0,5 2>/dev/null | grep -v amdgpu > $O/grid_shapes.txt
echo "grid shapes done"
for d in sparse ones half; do python3 bench.py --steps 10 --warmup 3 --no-cpu --no-configs --no-host-path --no-elim --density $d 2>/dev/null | grep "^{" >> $O/density.jsonl; done
AB_LIB=tools/libm4ri_hip_dev.so python3 tools/clock_probe.py 65536 --write > $O/clock.jsonl 2> $O/clock.err
cp profiles/clock_latest.json $O/clock_latest.json
echo "density / clock done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-configs --no-host-path --no-elim > $O/prof_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-configs --no-parity --no-host-path --no-elim > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-configs --no-parity --no-host-path --no-elim > $O/pmc_write.log 2>&1
echo "bench profiles done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lpn -- python3 $R/tools/lpn_pmc.py > $O/prof_lpn.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_lpn_fetch -- python3 $R/tools/lpn_pmc.py > $O/pmc_lpn_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_lpn_write -- python3 $R/tools/lpn_pmc.py > $O/pmc_lpn_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_configs -- python3 $R/tools/configs_only.py > $O/prof_configs.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_elim -- python3 $R/tools/elim_bench.py 4096 65536 --cpu-max 0 > $O/prof_elim.log 2>&1
echo "config / elimination profiles done"
cd $R
cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/final_kernel_stats.csv
cp $(ls $O/prof_lpn/*/*kernel_stats.csv | head -1) $O/lpn_kernel_stats.csv
cp $(ls $O/prof_configs/*/*kernel_stats.csv | head -1) $O/configs_kernel_stats.csv
cp $(ls $O/prof_elim/*/*kernel_stats.csv | head -1) $O/elim_kernel_stats.csv
python3 tools/pmc_summary.py FETCH_SIZE=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) WRITE_SIZE=$(ls $O/pmc_write/*/*counter_collection.csv | head -1) > $O/final_pmc_summary.json
python3 tools/pmc_summary.py FETCH_SIZE=$(ls $O/pmc_lpn_fetch/*/*counter_collection.csv | head -1) WRITE_SIZE=$(ls $O/pmc_lpn_write/*/*counter_collection.csv | head -1) > $O/lpn_pmc_summary.json
rm -rf $O/prof_elim $O/prof_bench $O/pmc_fetch $O/pmc_write $O/prof_lpn $O/pmc_lpn_fetch $O/pmc_lpn_write $O/prof_configs
python3 bench.py --gpus 2 --backend gloo --check --dim 16384 --no-cpu --steps 3 --warmup 1 2>$O/two_rank_rehearsal.err | grep "^{" > $O/two_rank_rehearsal.json
ls -la $O
