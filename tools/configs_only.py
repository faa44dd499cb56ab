#!/usr/bin/env python3
"""bench.py's `configs` leg alone, in a fresh process (development tool: separates the leg from what ran before it)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import m4ri_rust_amd  # noqa
from m4ri_rust_amd import device
comp = torch.cuda.Stream()
torch.cuda.set_stream(comp)
if os.environ.get("CONFIGS_NO_SHA"):
    bench._sha256_of = lambda dmat, stream: ""
vs = tuple(int(v) for v in os.environ.get("CONFIGS_LPN_V", "1,64,256").split(","))
for c in bench._extra_configs(device, torch, comp.cuda_stream, squares=not os.environ.get("CONFIGS_LPN_ONLY"), lpn_v=vs):
    print(c["workload"][:60], round(c["ms"] * 1e3, 2), "us", c["parity_sha256_ok"])
