"""What in a process slows `BinMatrix.transposed()` of a 65536^2 host matrix down? (development tool; result: freeing ~17 GiB of device
memory just before -- profiles/r04_host_path.txt).  python tools/ht_probe.py plain|torch|torchpin|streamsN|trim|main[_mul][_hash][_towords][_notrim][_sleep]|benchfn"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
if mode in ("torch", "torchpin"):
    import torch
    torch.cuda.init(); x = torch.zeros(10, device="cuda")
    if mode == "torchpin":
        h = torch.empty(256 << 20, dtype=torch.uint8).pin_memory(); d = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
        d.copy_(h, non_blocking=True); h.copy_(d, non_blocking=True); torch.cuda.synchronize(); del h, d
if mode.startswith("streams"):
    import torch
    keep = [torch.cuda.Stream() for _ in range(int(mode[7:]))]
    for st in keep:
        with torch.cuda.stream(st):
            y = torch.zeros(10, device="cuda") + 1
    torch.cuda.synchronize()
import m4ri_rust_amd as pkg
L = pkg._lib.lib()
if mode == "trim":
    B = pkg.BinMatrix.random(32768, 32768); C = B * B; del B, C
    L.gf2_trim()
if mode.startswith("main"):
    import torch, hashlib
    from m4ri_rust_amd import device as dev
    X, Y = dev.DMat.random(65536, 65536, 1), dev.DMat.random(65536, 65536, 2)
    Z = dev.DMat(65536, 65536)
    if "mul" in mode:
        for _ in range(3): dev.mul(X, Y, C=Z)
        torch.cuda.synchronize()
    if "hash" in mode:
        hashlib.sha256(Z.to_words().tobytes()).hexdigest()
    if "towords" in mode:
        w = Z.to_words(); del w
    del X, Y, Z
    torch.cuda.empty_cache()
    if "notrim" not in mode: L.gf2_trim()
    if "sleep" in mode: time.sleep(3)
A = pkg.BinMatrix.random(65536, 65536)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); T = A.transposed(); ts.append(time.perf_counter() - t0); del T
print(mode, " ".join("%.2f" % (t * 1e3) for t in ts), flush=True)
if mode == "benchfn":
    import torch, json
    sys.path.insert(0, ROOT)
    import bench
    r = bench._host_path(torch, sizes=())
    print(json.dumps(r["entries"][0])[:200])
