"""profiles/traffic_latest.json from a pmc summary of the bench command (tools/pmc_summary.py output):
     python tools/make_traffic_json.py profiles/r04_final_pmc_summary.json > profiles/traffic_latest.json
HBM bytes of ONE PRODUCT's tile launches = FETCH_SIZE x 2 (gfx950 tallies 128-byte read requests at 64 bytes, MI355X_MICROARCH.md,
HBM section; calibrated in profiles/r01_fetch_size_calibration.txt) + WRITE_SIZE, summed over every tile-kernel dispatch of the
product (the main v8 launch, the tail launch of the last leaves, the stream-K reduction): bench.py's event pair brackets exactly
those, so `roofline.traffic`, `roofline.achieved` and `avg_launch_ms` refer to the same set of kernels."""
import json
import sys

s = json.load(open(sys.argv[1]))
names = [k for k in s["FETCH_SIZE"] if "gf2_m4rm_kernel" in k or "gf2_streamk_reduce_kernel" in k or "gf2_splitk_reduce_kernel" in k]
main = max((k for k in names if "gf2_m4rm_kernel_v8<8" in k), key=lambda k: s["FETCH_SIZE"][k]["avg_KB_per_dispatch"])
products = s["FETCH_SIZE"][main]["dispatches"]  # one main launch per product
f = sum(s["FETCH_SIZE"][k]["avg_KB_per_dispatch"] * s["FETCH_SIZE"][k]["dispatches"] for k in names) / products
w = sum(s["WRITE_SIZE"][k]["avg_KB_per_dispatch"] * s["WRITE_SIZE"][k]["dispatches"] for k in names if k in s["WRITE_SIZE"]) / products
json.dump({
    "n": 65536, "levels": 4, "n_gpus": 1, "kernels": names, "main_kernel": main, "products": products,
    "FETCH_SIZE_KB_raw_per_product": f, "WRITE_SIZE_KB_raw_per_product": w,
    "correction": "gfx950: FETCH_SIZE tallies 128-B read requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section; calibrated on this "
                  "repo's kernels in profiles/r01_fetch_size_calibration.txt); WRITE_SIZE is exact",
    "hbm_bytes_per_launch": (2 * f + w) * 1024.0, "launches_per_product": 1,
    "covers": "all tile-kernel dispatches of one product (main launch + tail launch + stream-K reduction) = what bench.py's event pair brackets",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only, python3 bench.py --steps 2 "
              "--warmup 1 --no-cpu --no-configs --no-parity --no-host-path (%s, tools/pmc_summary.py, tools/collect_profiles_r04.sh)" % sys.argv[1],
}, sys.stdout, indent=1)
print()
