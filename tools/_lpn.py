import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from m4ri_rust_amd import device as dev
for v in (64, 128, 256):
    A, B, C = dev.DMat.random(1 << 20, 256, 1), dev.DMat.random(256, v, 2), dev.DMat(1 << 20, v)
    for _ in range(20):
        dev.mul(A, B, C, algo="m4rm")
    torch.cuda.synchronize()
