"""A few 65536^2 transpositions for counter runs (development tool): rocprofv3 --pmc ... -- python3 tools/transpose_pmc.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from m4ri_rust_amd import device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
S, D = dev.DMat.random(n, n, 1), dev.DMat(n, n)
for _ in range(4):
    dev.transpose(S, D)
torch.cuda.synchronize()
