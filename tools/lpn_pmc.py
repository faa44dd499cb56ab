"""A few LPN-shaped products (BASELINE config 5) for counter runs:
   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o f -- python3 tools/lpn_pmc.py   (then WRITE_SIZE in a second run)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from m4ri_rust_amd import device as dev
for v in (1, 64, 128, 256):
    A, B, C = dev.DMat.random(1 << 20, 256, 1), dev.DMat.random(256, v, 2), dev.DMat(1 << 20, v)
    for _ in range(4):
        dev.mul(A, B, C, algo="naive")
    torch.cuda.synchronize()
