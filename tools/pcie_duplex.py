#!/usr/bin/env python3
"""PCIe rates of the box: host-to-device alone, device-to-host alone, both at once on two streams (development tool)."""
import time
import torch
n = 256 << 20
h1 = torch.empty(n, dtype=torch.uint8).pin_memory()
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d1 = torch.empty(n, dtype=torch.uint8, device="cuda")
d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(up, down, reps=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if up:
            with torch.cuda.stream(s1):
                d1.copy_(h1, non_blocking=True)
        if down:
            with torch.cuda.stream(s2):
                h2.copy_(d2, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for _ in range(2):
    run(True, True)
tu, td, tb = run(True, False), run(False, True), run(True, True)
print("H2D %.1f GB/s  D2H %.1f GB/s  both at once: %.2f ms for %d MiB each way = %.1f GB/s per direction" % (
    n / tu / 1e9, n / td / 1e9, tb * 1e3, n >> 20, n / tb / 1e9))
