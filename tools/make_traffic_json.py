"""profiles/traffic_latest.json from a pmc summary of the bench command (tools/pmc_summary.py output):
     python tools/make_traffic_json.py profiles/r03_final_pmc_summary.json > profiles/traffic_latest.json
HBM bytes of the dominant kernel's launch = FETCH_SIZE x 2 (gfx950 tallies 128-byte read requests at 64 bytes, MI355X_MICROARCH.md,
HBM section; calibrated in profiles/r01_fetch_size_calibration.txt) + WRITE_SIZE."""
import json
import sys

s = json.load(open(sys.argv[1]))
name = max((k for k in s["FETCH_SIZE"] if "gf2_m4rm_kernel_v8<8" in k), key=lambda k: s["FETCH_SIZE"][k]["avg_KB_per_dispatch"])
f, w = s["FETCH_SIZE"][name]["avg_KB_per_dispatch"], s["WRITE_SIZE"][name]["avg_KB_per_dispatch"]
json.dump({
    "n": 65536, "levels": 4, "n_gpus": 1, "kernel": name,
    "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w,
    "correction": "gfx950: FETCH_SIZE tallies 128-B read requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section; calibrated on this "
                  "repo's kernels in profiles/r01_fetch_size_calibration.txt); WRITE_SIZE is exact",
    "hbm_bytes_per_launch": (2 * f + w) * 1024.0,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only, python3 bench.py --steps 2 "
              "--warmup 1 --no-cpu --no-configs --no-parity, round 3 (%s, tools/pmc_summary.py, tools/collect_profiles.sh)" % sys.argv[1],
}, sys.stdout, indent=1)
print()
