import torch, time
for gib in (1, 4):
    t = torch.empty(gib << 27, dtype=torch.int64, device="cuda")
    s = torch.empty(gib << 27, dtype=torch.int64, device="cuda")
    for name, fn, bytes_ in (("fill (write only)", lambda: t.fill_(1), gib << 30), ("copy (read + write)", lambda: t.copy_(s), 2 * (gib << 30)),
                             ("sum (read only)", lambda: s.sum(), gib << 30)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("%d GiB %-22s %.3f ms  %.0f GB/s" % (gib, name, dt * 1e3, bytes_ / dt / 1e9))
