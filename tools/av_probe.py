"""`&A * &v` (2^20 x 256) on host operands with A uploaded by every call: the pipeline's block count against the raw copy rates
of the box (development tool; one subprocess per M4RI_HIP_HOST_PIPELINE_BLOCKS value, the library reads it once)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import m4ri_rust_amd as pkg
    L = pkg._lib.lib()
    m, l = 1 << 20, 256
    A = pkg.BinMatrix.random(m, l)
    vrow = pkg.BinMatrix.random(1, l)
    def op():
        vt = L.mzd_transpose(None, vrow.mzd)
        r = L.mzd_mul_naive(None, A.mzd, vt)
        rt = L.mzd_transpose(None, r)
        L.mzd_free(vt), L.mzd_free(r), L.mzd_free(rt)
    for cached in (False, True):
        if cached: L.gf2_mzd_cache_on_device(A.mzd)
        op(); op()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); op(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print("  blocks=%s %s: min %.1f us, median %.1f us" % (os.environ.get("M4RI_HIP_HOST_PIPELINE_BLOCKS", "default"),
              "A cached" if cached else "A uploaded", ts[0] * 1e6, ts[len(ts) // 2] * 1e6))
    sys.exit(0)
import torch
for nb in (8 << 20, 32 << 20):
    h = torch.empty(nb, dtype=torch.uint8).pin_memory(); d = torch.empty(nb, dtype=torch.uint8, device="cuda")
    for name, fn in (("h2d", lambda: d.copy_(h, non_blocking=True)), ("d2h", lambda: h.copy_(d, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print("raw %s of %d MiB (pinned, one copy + sync): min %.1f us" % (name, nb >> 20, min(ts) * 1e6))
for b in ("0", "2", "4", "8", "16"):
    env = dict(os.environ, M4RI_HIP_HOST_PIPELINE_BLOCKS=b)
    print(subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True).stdout, end="")
