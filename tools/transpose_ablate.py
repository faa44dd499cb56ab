"""Timing of dev.transpose at 65536^2 under the development library's M4RI_HIP_TRANSPOSE_FLAGS (timing-only ablations give wrong results):
     AB_LIB=tools/libm4ri_hip_dev.so M4RI_HIP_TRANSPOSE_FLAGS=<f> python tools/transpose_ablate.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("AB_LIB"):
    import m4ri_rust_amd  # noqa
    from m4ri_rust_amd import _lib
    _lib.LIB_PATH = os.environ["AB_LIB"]
import torch
from m4ri_rust_amd import device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
S, D = dev.DMat.random(n, n, 1), dev.DMat(n, n)
for _ in range(3):
    dev.transpose(S, D)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(10):
        dev.transpose(S, D)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 10)
print("flags=%s n=%d: %.4f ms" % (os.environ.get("M4RI_HIP_TRANSPOSE_FLAGS", "default"), n, best * 1e3), flush=True)
