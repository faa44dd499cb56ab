import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from m4ri_rust_amd import device as dev
for n in (2048, 4096, 8192):
    A, B, C = dev.DMat.random(n, n, 1), dev.DMat.random(n, n, 2), dev.DMat(n, n)
    for _ in range(20):
        dev.mul(A, B, C, algo="m4rm")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        dev.mul(A, B, C, algo="m4rm")
    torch.cuda.synchronize()
    print("ksplit %s n=%d: %.1f us" % (os.environ.get("M4RI_HIP_M4RM_KSPLIT", "auto"), n, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
