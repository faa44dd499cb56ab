#!/usr/bin/env python3
"""Where does a fused elimination step's time go?  (development tool: AB_LIB=tools/libm4ri_hip_dev.so python3 tools/elim_stamps.py [n ...])
The look-ahead workgroup of gf2_elim_update_kernel<true> adds the shader cycles of its stages to a stamp buffer (gf2_elim.hip, ELIM_STAMP)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import m4ri_rust_amd as pkg
from m4ri_rust_amd import _lib, device
if os.environ.get("AB_LIB"):
    _lib.LIB_PATH = os.environ["AB_LIB"]
L = _lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
if not hasattr(raw, "gf2k_dev_set_elim_stamps"):
    sys.exit("elim_stamps: needs the development library (AB_LIB=tools/libm4ri_hip_dev.so)")
raw.gf2k_dev_set_elim_stamps.argtypes = [ctypes.c_void_p]
names = ["", "launch start -> search starts (the stash: no wait any more)", "candidates of the first pass loaded", "basis complete (search loop)",
         "every update workgroup done (cnt2)", "published (flags, state, raw pivot rows, selector map)"]
for n in [int(a) for a in sys.argv[1:]] or [4096, 65536]:
    M = device.DMat.random(n, n, 5)
    device.echelonize(M, full=True)
    st = torch.zeros(32, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert raw.gf2k_dev_set_elim_stamps(st.data_ptr()) == 0
    M = device.DMat.random(n, n, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    device.echelonize(M, full=True)
    dt = time.perf_counter() - t0
    assert raw.gf2k_dev_set_elim_stamps(None) == 0
    h = st.cpu().numpy()
    steps = int(h[8])
    print("n = %d: %.2f ms, %d look-ahead steps; shader cycles per step (at ~2.1 GHz: 2100 cycles = 1 us)" % (n, dt * 1e3, steps))
    tot = 0
    for k in range(1, 6):
        print("  %-62s %8.0f" % (names[k], h[k] / max(steps, 1)))
        tot += h[k]
    for k, nm in ((7, "re-reduction (wave 0)"), (9, "barrier behind the re-reductions"), (10, "insertions by wave 0"), (11, "barrier behind another wave's insertions")):
        print("    of the search loop: %-42s %8.0f" % (nm, h[k] / max(steps, 1)))
        tot += h[k]
    print("  %-62s %8.0f  (= %.1f us at 2.1 GHz; the launch itself adds its boundary)" % ("sum", tot / max(steps, 1), tot / max(steps, 1) / 2100.0))
    print("  the middle UPDATE workgroup of the same launches (thread 0's view):")
    tot = 0
    for k, nm in ((3, "start -> tables built"), (4, "wave 0's run of rows done (stores issued)"), (5, "every wave done, stores landed, cnt2 raised")):
        print("    %-60s %8.0f" % (nm, h[16 + k] / max(steps, 1)))
        tot += h[16 + k]
    print("    %-60s %8.0f" % ("sum", tot / max(steps, 1)))
