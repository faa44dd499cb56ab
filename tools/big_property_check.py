"""(AB)x == A(Bx) on the device at 131072^3 (5 Strassen levels, 141 GiB arena) and 262144^3 (levels capped by memory): the
size-independent check behind the large-size numbers in DESIGN.md (development tool; needs most of the 288 GB)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import m4ri_rust_amd as pkg
from m4ri_rust_amd import device as dev
for n in (131072, 262144):
    A, B = dev.DMat.random(n, n, 1), dev.DMat.random(n, n, 2)
    x = dev.DMat.random(n, 64, 3)
    t = time.time(); P = dev.mul(A, B); torch.cuda.synchronize(); dt = time.time() - t
    lhs = dev.mul(P, x, algo="m4rm")
    rhs = dev.mul(A, dev.mul(B, x, algo="m4rm"), algo="m4rm")
    ok = dev.equal(lhs, rhs)
    print(n, "levels", pkg._lib.lib().gf2_strassen_levels(n, n, n, 0, 0), "first product %.3f s" % dt, "(AB)x == A(Bx):", ok, flush=True)
    assert ok
    del A, B, P, x, lhs, rhs
    pkg._lib.lib().gf2_trim()
print("OK")
