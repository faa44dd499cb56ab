#!/usr/bin/env python3
"""Pieces of the 2 x 2 host plan at 32768^3 (development tool): the quarter products on strided device views, linear and 2-D
PCIe copies of the operand pieces, and the whole call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import m4ri_rust_amd as pkg
from m4ri_rust_amd import device
device.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
w = n // 64
A_t = torch.randint(-2**62, 2**62, (n, w), dtype=torch.int64, device="cuda")
B_t = torch.randint(-2**62, 2**62, (n, w), dtype=torch.int64, device="cuda")
C_t = torch.empty((n, w), dtype=torch.int64, device="cuda")
def view(t, r0, rows, c0w, cols):
    return device.DMat.wrap(t.data_ptr() + 8 * (r0 * w + c0w), rows, cols, w, keep=t)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
h = n // 2
print("whole %d^3 resident: %.2f ms" % (n, timeit(lambda: device.mul(view(A_t, 0, n, 0, n), view(B_t, 0, n, 0, n), C=view(C_t, 0, n, 0, n)))))
for (i, j) in ((0, 0), (1, 1)):
    a, b, c = view(A_t, i * h, h, 0, n), view(B_t, 0, n, j * (h // 64), h), view(C_t, i * h, h, j * (h // 64), h)
    print("quarter (%d,%d) %dx%dx%d on strided views: %.2f ms" % (i, j, h, n, h, timeit(lambda: device.mul(a, b, C=c))))
a, b, c = view(A_t, 0, h, 0, n), view(B_t, 0, n, 0, n), view(C_t, 0, h, 0, n)
print("row half %dx%dx%d: %.2f ms" % (h, n, n, timeit(lambda: device.mul(a, b, C=c))))
a, b, c = view(A_t, 0, n // 4, 0, n), view(B_t, 0, n, 0, n), view(C_t, 0, n // 4, 0, n)
print("row quarter %dx%dx%d: %.2f ms" % (n // 4, n, n, timeit(lambda: device.mul(a, b, C=c))))
# PCIe: linear half of A, 2-D column half of B, 2-D quadrant of C
hp = torch.empty((n, w), dtype=torch.int64).pin_memory()
s = torch.cuda.Stream()
def cp(dst, src):
    with torch.cuda.stream(s):
        dst.copy_(src, non_blocking=True)
    s.synchronize()
for name, dst, src in (("H2D linear half (rows)", A_t[:h], hp[:h]), ("H2D 2-D column half", B_t[:, :w // 2], hp[:, :w // 2]),
                       ("D2H 2-D quadrant", hp[:h, :w // 2], C_t[:h, :w // 2]), ("D2H linear quarter rows", hp[:n // 4], C_t[:n // 4])):
    cp(dst, src)
    t0 = time.perf_counter()
    for _ in range(3): cp(dst, src)
    dt = (time.perf_counter() - t0) / 3
    nbytes = dst.numel() * 8
    print("%-26s %6.1f MiB in %.2f ms = %.1f GB/s" % (name, nbytes / 2**20, dt * 1e3, nbytes / dt / 1e9))
