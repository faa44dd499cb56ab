#!/usr/bin/env python3
"""bench.py -- GF(2) n x n x n matrix product on MI355X, BASELINE.json's headline metric.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full product C = A*B over GF(2) with A and B already resident in HBM
(n = 65536 by default; it fits one GPU: 3 x 512 MiB + workspace).  With N > 1 ranks
(launched by torch.distributed.run, one process per GPU, RCCL) the total work is fixed
(strong scaling): rank r keeps row block r of A resident, B lives on rank 0, and every step
does  broadcast(B)  ->  local product of the row block  ->  gather(C blocks) on rank 0.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on its own
stream inside the library) and `cpu_baseline` (the oracle's single-thread M4RM+Strassen port on
a bounded sample, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0      # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy rate
HBM_COPY_GBS = 6290.0
LDS_PEAK_TBS = 150.0       # aggregate ds_read_b128 rate, all CUs (MI355X_MICROARCH.md, LDS)
LDS_BYTES_PER_CLK = 256 * 256  # 256 B per clock and CU (64 dwords wide), 256 CUs (MI355X_MICROARCH.md, LDS)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _launch_ranks(n):
    import socket
    import subprocess
    with socket.socket() as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def _cpu_baseline(args):
    """The reported CPU baseline, run BEFORE the GPU loop (rank 0, N = 1): oracle_mul_fast, the single-thread port of the
    reference's algorithm (M4RI itself is absent), on one bounded sample product; plus a threaded figure, labelled
    non-reference.  Returns (json object, sample product words)."""
    import numpy as np
    import gf2util as g
    cn = args.cpu_n
    a, b = g.random_words(cn, cn, 1), g.random_words(cn, cn, 2)
    g.oracle()
    t1 = time.perf_counter()
    c = g.o_mul_fast(a, b, cn, cn, cn)
    cdt = time.perf_counter() - t1
    out = {
        "value": 2.0 * cn ** 3 / cdt, "unit": "bit-ops/s", "cores": 1, "kind": "port",
        "sample": "one %dx%dx%d product by oracle_mul_fast (single-thread M4RM k=8 + Strassen-Winograd, "
                  "gcc -Ofast, no -march), %.2f s; M4RI itself is absent from the reference tree" % (cn, cn, cn, cdt),
        "host_cpus": os.cpu_count(), "host_cpu_model": _cpu_model(),
    }
    # optional second figure: the same port on several cores, row blocks of A in threads
    # (ctypes releases the GIL).  NOT the reference's configuration: its M4RI build is single-threaded.
    nthr = min(args.cpu_threads, os.cpu_count() or 1, cn // 2048)
    if nthr > 1:
        import threading
        blk = cn // nthr
        outs = [None] * nthr

        def work(k):
            outs[k] = g.o_mul_fast(np.ascontiguousarray(a[k * blk:(k + 1) * blk]), b, blk, cn, cn)

        t1 = time.perf_counter()
        ths = [threading.Thread(target=work, args=(k,)) for k in range(nthr)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        mdt = time.perf_counter() - t1
        out["threaded_port"] = {
            "value": 2.0 * (blk * nthr) * cn * cn / mdt, "unit": "bit-ops/s", "cores": nthr,
            "matches_single_core": bool(np.array_equal(np.concatenate(outs), c[:blk * nthr])),
            "note": "non-reference configuration (the reference builds M4RI single-threaded): row blocks of A in %d threads, %.2f s" % (nthr, mdt)}
    return out, c


def _golden_digests():
    """Committed full-size digests (tests/golden/digests_large.json, written by tests/golden/make_golden_large.py): data, not
    the oracle -- the bench compares the products it has just timed with them."""
    out = {}
    for name in ("digests.json", "digests_large.json"):
        try:
            with open(os.path.join(ROOT, "tests", "golden", name)) as f:
                out.update(json.load(f))
        except (OSError, ValueError):
            pass
    return out


def _sha256_of(dmat, stream):
    import hashlib
    return hashlib.sha256(dmat.to_words(stream).tobytes()).hexdigest()


def _timed(fn, reps, torch):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def _timed_median(fn, reps, torch, segments=5):
    """Microsecond-scale products: `segments` back-to-back timed runs of `reps` products each, the median of their averages (one
    host hiccup of a millisecond inside a 6-ms loop once read as 42.9 us per 4096^3 product instead of 31.9)."""
    ts = sorted(_timed(fn, reps, torch) for _ in range(segments))
    return ts[len(ts) // 2]


def _extra_configs(device, torch, stream, squares=True, lpn_v=(1, 64, 128, 256)):
    """The other single-GPU BASELINE.json configurations, timed in the same run as the headline (device resident, wall clock
    around `reps` back-to-back products on the bench stream) and hashed against the committed digests:
      config 2  4096^3, M4RM kernel only;  config 3  32768^3, Strassen over M4RM;
      config 5  2^20 x 256 times 256 x V, V = 1 / 64 / 128 / 256, COLD: ten A (and C) buffers visited round-robin, 320 MiB of A
                between two uses of the same buffer -- more than the 256 MiB Infinity Cache holds."""
    dig = _golden_digests()
    out = []

    def entry(name, m, l, n, algo, dt, sha, key, extra=None):
        wl, wn = (l + 63) // 64, (n + 63) // 64
        layout = 8.0 * (m * wl + l * wn + m * wn)  # operands in the M4RI layout (rows padded to 64-bit words), each moved once
        # algorithmic bytes = (m l + l n + m n) / 8 with ROWS OF A AND B padded to words but C counted by its
        # bits (2^20 x 256 times 256 x 1: 32 MiB + 32 B + 2^20 / 8 B = 33.7 MB); the M4RI layout stores that C as one 64-bit word
        # per row (41.9 MB moved).  `frac` uses the section 8(d) figure; the layout figure is kept beside it
        alg = 8.0 * m * wl + l * n / 8.0 + m * n / 8.0
        e = {"workload": name, "m": m, "l": l, "n": n, "algo": algo, "ms": dt * 1e3, "bit_ops_per_s": 2.0 * m * l * n / dt,
             "roofline": {"bound": "hbm", "achieved": alg / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": alg / dt / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg,
                          "layout_bytes": layout, "layout_frac": layout / dt / 1e9 / HBM_PEAK_GBS},
             "strassen_levels": device._lib.lib().gf2_strassen_levels(m, l, n, device.ALGOS[algo], 0)}
        if extra:
            e.update(extra)
        want = dig.get(key, {}).get("sha256_c")
        e["parity_sha256_ok"] = (sha == want) if want else None
        out.append(e)

    def onchip(m, l, n, dt):  # every 8 bits of the inner dimension cost one 16-byte LDS read per 128 columns (M4RM, k = 8)
        lds = m * (l / 8.0) * (n / 128.0) * 16.0
        return {"roofline_onchip": {"bound": "lds", "achieved": lds / dt / 1e12, "peak": LDS_PEAK_TBS, "unit": "TB/s",
                                    "frac": lds / dt / 1e12 / LDS_PEAK_TBS,
                                    "note": "plain M4RM read count; Strassen levels lower the reads actually made"}}

    for name, nn, algo, reps, key in ((("config 2: 4096^3, M4RM kernel only", 4096, "m4rm", 200, "sq_4096"),
                                       ("config 3: 32768^3, Strassen over M4RM", 32768, "auto", 10, "sq_32768")) if squares else ()):
        A, B, C = device.DMat.random(nn, nn, 1, stream), device.DMat.random(nn, nn, 2, stream), device.DMat(nn, nn)
        # untimed: ~30 ms of the same product first.  The clocks drop during the host-side hashing between two configurations and
        # take more than 2 ms of work to come back (measured on the LPN shapes below: 23.9 us after 100 untimed products, 20.3 us
        # after 1000, for the same 200 timed ones)
        for _ in range(1000 if nn <= 4096 else 8):
            device.mul(A, B, C=C, algo=algo, stream=stream)
        dt = (_timed_median if nn <= 4096 else _timed)(lambda i: device.mul(A, B, C=C, algo=algo, stream=stream), reps, torch)
        entry(name, nn, nn, nn, algo, dt, _sha256_of(C, stream), key, onchip(nn, nn, nn, dt))
        del A, B, C
    m, l, nbuf = 1 << 20, 256, 10
    # The ten A buffers lie end to end in one allocation, the ten C buffers in another with 3 MiB + 68 KiB more between
    # neighbours than they need, and A_i is paired with C_(3i+1 mod nbuf): the distance between a product's read and its write
    # stream decides how the two meet in the memory system (tools/lpn_placement.py, profiles/r04_lpn_placement.txt: V = 128 takes
    # 11.1-11.5 us at most distances and 12.1-12.8 when the distance is within half a MiB of a multiple of 64 MiB; the round-3
    # kernel for V = 256 took 18.9-23.0 us depending on it alone).  Separate allocations of equal size have distances that all
    # fall into ONE residue class -- a lucky or an unlucky one, per process --; ten different residues make the figure typical.
    MiB = 1 << 20
    a_slab = torch.empty(nbuf * m * (l // 64), dtype=torch.int64, device="cuda")
    As = [device.DMat.wrap(a_slab.data_ptr() + i * m * (l // 8), m, l, l // 64, keep=a_slab) for i in range(nbuf)]
    for i, a in enumerate(As):
        a.fill_random(1 if i == 0 else 100 + i, stream)
    c_pitch = m * 32 + 3 * MiB + 68 * 1024
    c_slab = torch.empty(nbuf * c_pitch // 8, dtype=torch.int64, device="cuda")
    for V in lpn_v:
        X = device.DMat.random(l, V, 2, stream)
        wv = (V + 63) // 64
        Cs = [device.DMat.wrap(c_slab.data_ptr() + j * c_pitch, m, V, wv, keep=c_slab) for j in range(nbuf)]

        def one(i):
            device.mul(As[i % nbuf], X, C=Cs[(3 * i + 1) % nbuf], algo="naive", stream=stream)
        for i in range(100 * nbuf):  # untimed round-robin passes, 10-25 ms (see above)
            one(i)
        dt = _timed_median(one, 20 * nbuf, torch)
        entry("config 5: LPN 2^20 x 256 times 256 x %d (mzd_mul_naive entry), cold: %d rotating A buffers" % (V, nbuf),
              m, l, V, "naive", dt, _sha256_of(Cs[1], stream), "lpn_1048576x256x%d" % V)  # Cs[1] = A_0 * X
        del Cs, X
    del As, a_slab, c_slab
    return out


def _host_path(torch, sizes=(32768, 65536)):
    """SURVEY.md section 8(d) "wall-clock end-to-end through the C ABI incl. H2D/D2H": what a Rust caller of the drop-in library
    gets on HOST mzd_t operands (m4ri-rust/src/friendly/binary_matrix.rs:459-472 `&A * &B`, :528-542 `&A * &v`, :272-279
    `transposed()`), each beside its PCIe floor = bytes that must cross the link / the rate this box reaches in that direction
    (measured here on pinned buffers; uploads and downloads overlap, so the floor is the larger of the two directions).
    Never `value`: the headline is quoted on resident operands."""
    import m4ri_rust_amd as pkg
    L = pkg._lib.lib()
    out = []
    # PCIe rates of this box, pinned host memory, 256 MiB each way
    nb = 256 << 20
    h = torch.empty(nb, dtype=torch.uint8).pin_memory()
    d = torch.empty(nb, dtype=torch.uint8, device="cuda")
    rates = {}

    def measure():
        r = {}
        for name, fn in (("h2d", lambda: d.copy_(h, non_blocking=True)), ("d2h", lambda: h.copy_(d, non_blocking=True))):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
            r[name] = 4 * nb / (time.perf_counter() - t0)
        return r
    # The process has just handed ~20 GiB of device memory back to the driver, which keeps the copy engines busy for a while
    # (a third of the copy rate, or less, for 2-3 s: tools/ht_probe.py, profiles/r04_host_path.txt): measure until two consecutive
    # rates agree within 3 % (at most ~10 s), so that neither the floors nor the legs below carry the free.
    settle = 0
    for _ in range(40):
        r = measure()
        if rates and all(abs(r[k] - rates[k]) <= 0.03 * rates[k] for k in r):
            rates = {k: max(r[k], rates[k]) for k in r}
            break
        rates = r
        settle += 1
        time.sleep(0.25)
    del h, d

    def floor_ms(up, down):
        return max(up / rates["h2d"], down / rates["d2h"]) * 1e3

    def best(fn, reps=3):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts), ts[0]

    # transposed() first: the first 512 MiB destination of this process has to be pinned (nothing of that size is in the pool yet)
    n = 65536
    mb = n * n / 8.0
    A = pkg.BinMatrix.random(n, n)
    t0 = time.perf_counter()
    T = A.transposed()
    t_first = time.perf_counter() - t0
    del T  # its block goes into the pool: later destinations of this size find it there
    t_warm, _ = best(lambda: A.transposed())
    out.append({"workload": "mzd_transpose(NULL, A) on host mzd_t, %d^2" % n, "ms": t_warm * 1e3, "first_call_ms": t_first * 1e3,
                "pcie_floor_ms": floor_ms(mb, mb), "bytes_up": mb, "bytes_down": mb,
                "note": "first_call_ms: the destination's 512 MiB block is pinned inside the call (hipHostMalloc); later calls find a "
                        "pooled block; gf2_mzd_prewarm(r, c, count) moves that cost out of the first call"})
    del A
    for n in sizes:
        A, B = pkg.BinMatrix.random(n, n), pkg.BinMatrix.random(n, n)
        mb = n * n / 8.0

        def mul_null():
            L.mzd_free(L.mzd_mul(None, A.mzd, B.mzd, 0))
        first = time.perf_counter()
        mul_null()  # the first product of this size in the process: grows the device arenas (and pins C's block if none is pooled)
        first = time.perf_counter() - first
        t_null, _ = best(mul_null)
        Cp = pkg.BinMatrix.zero(n, n)
        t_pre, _ = best(lambda: L.mzd_mul(Cp.mzd, A.mzd, B.mzd, 0))
        out.append({"workload": "mzd_mul(NULL, A, B, 0) on host mzd_t, %d^3 (upload A, B; product; download C)" % n, "ms": t_null * 1e3,
                    "first_call_ms": first * 1e3, "preallocated_c_ms": t_pre * 1e3, "pcie_floor_ms": floor_ms(2 * mb, mb),
                    "bytes_up": 2 * mb, "bytes_down": mb, "bit_ops_per_s": 2.0 * n ** 3 / t_pre})
        del Cp, A, B
    # `&A * &v` with 2^20 LPN samples of 256 bits, as the C calls the Rust operator makes (binary_matrix.rs:416-431 mul_slice:
    # from_slices -> transposed -> mzd_mul_naive(NULL, A, v^T); :528-542 as_vector: transposed): A uploaded on every call, and
    # kept on the device (gf2_mzd_cache_on_device)
    m, l = 1 << 20, 256
    A = pkg.BinMatrix.random(m, l)
    vrow = pkg.BinMatrix.random(1, l)

    def a_times_v():
        vt = L.mzd_transpose(None, vrow.mzd)            # l x 1
        r = L.mzd_mul_naive(None, A.mzd, vt)             # m x 1
        rt = L.mzd_transpose(None, r)                    # 1 x m: what as_vector copies out of
        L.mzd_free(vt), L.mzd_free(r), L.mzd_free(rt)
    a_times_v(), a_times_v()
    t_unc, _ = best(a_times_v, 9)  # (sub-millisecond calls: the best of nine, two untimed ones first)
    L.gf2_mzd_cache_on_device(A.mzd)
    a_times_v(), a_times_v()
    t_c, _ = best(a_times_v, 9)
    L.gf2_mzd_uncache(A.mzd)
    out.append({"workload": "&A * &v on host operands, 2^20 x 256: mzd_transpose(v), mzd_mul_naive(NULL, A, v^T), mzd_transpose(result); "
                            "A uploaded per call", "ms": t_unc * 1e3,
                "pcie_floor_ms": floor_ms(m * l / 8.0, m * 8.0), "bytes_up": m * l / 8.0, "bytes_down": m * 8.0})
    out.append({"workload": "the same with gf2_mzd_cache_on_device(A)", "ms": t_c * 1e3, "pcie_floor_ms": floor_ms(64.0, m * 8.0),
                "bytes_up": 64.0, "bytes_down": m * 8.0,
                "note": "C of a matrix x vector product is one 64-bit word per row in the M4RI layout: 8 MiB come back for 128 KiB of bits"})
    return {"pcie_GBps": {k: v / 1e9 for k, v in rates.items()}, "pcie_settle_rounds": settle, "entries": out}


def _elimination(device, torch, sizes=(4096, 65536)):
    """SURVEY.md section 8(f) row 3 in the driver line: reduced echelon form (mzd_echelonize(A, 1), echelonform.rs:16) of a random
    n x n matrix, device resident (gf2_echelonize_dev), best of three; the blocked Gauss-Jordan of DESIGN.md section 7.1.  A random
    square matrix has rank n - d with probability ~0.29 / 0.58 / 0.13 for d = 0 / 1 / 2: the rank is reported, the bits are the
    business of tests/test_gpu_elim.py (oracle, fixtures, E * A = R at full size)."""
    out = []
    for n in sizes:
        ts, rank = [], None
        for _ in range(3):
            M = device.DMat.random(n, n, 5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rank, _ = device.echelonize(M, full=True)
            ts.append(time.perf_counter() - t0)
            del M
        out.append({"workload": "reduced echelon form of a random %d x %d matrix (seed 5), device resident" % (n, n), "n": n, "ms": min(ts) * 1e3,
                    "rank": int(rank), "n3_per_s": float(n) ** 3 / min(ts)})
    return out


def _sharded_self_check(device, sharded, torch, Bp, Cfull_t, n, nblocks, stream):
    """Cheap default check of a multi-GPU step (rank 0): eight consecutive rows out of every ROW BLOCK of the grid (at least 32
    rows in all, offsets from a fixed seed) are re-multiplied by rank 0 alone from the seeded generator, against every (sub-)panel
    of B, and compared with the rows of C the ranks produced and RCCL gathered -- so every rank's block of C is sampled."""
    import random
    rng = random.Random(20261004)
    rows = n // nblocks
    grp = max(8, -(-32 // nblocks))
    starts = [r * rows + rng.randrange(0, rows - grp + 1) for r in range(nblocks)]
    As_t = torch.empty((grp * nblocks, n // 64), dtype=torch.int64, device="cuda")
    for k, r0 in enumerate(starts):
        sharded.fill_row_block(device.DMat.from_torch(As_t[k * grp:(k + 1) * grp], n), seed=1, row0=r0, stream=stream)
    idx = torch.tensor([r0 + j for r0 in starts for j in range(grp)], device="cuda")
    As = device.DMat.from_torch(As_t, n)
    import numpy as np
    ok = True
    for pnl in range(len(Bp)):
        ref = device.mul(As, Bp[pnl], algo="m4rm", stream=stream).to_words(stream)
        got = Cfull_t[pnl][idx].cpu().numpy().view(np.uint64)
        ok = ok and bool(np.array_equal(got, ref))
    return {"rows_checked": int(idx.numel()), "panels_checked": len(Bp), "row_starts": starts, "ok": bool(ok)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dim", "--n", dest="n", type=int, default=65536, help="matrix dimension (BASELINE metric: 65536)")
    ap.add_argument("--algo", default="auto", choices=["auto", "m4rm", "strassen"])
    ap.add_argument("--levels", type=int, default=0, help="Strassen levels (0 = automatic)")
    ap.add_argument("--cpu-n", type=int, default=32768, help="dimension of the CPU-baseline sample product")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="threads of the additional multi-core CPU figure (0/1 = skip; 16 = a one-GPU box's CPU share)")
    ap.add_argument("--panels", type=int, default=0,
                    help="column panels of B per step when N > 1 (RCCL/compute overlap); 0 = 2 panels on 2 GPUs, 4 above "
                         "(measured per-rank products: thin panels cost Strassen efficiency, 32768x65536x32768 takes 13.0 ms "
                         "but 4 x 32768x65536x16384 take 31 ms; from 4 GPUs on the transfer is the longer leg)")
    ap.add_argument("--grid", default=None,
                    help="N > 1: RxQ grid of ranks (R * Q = N): rank i * Q + j keeps row block i of A (n / R rows) and receives only column "
                         "panel j of B (n / Q columns); Nx1 is the plain row-block scheme.  Unset: every factorisation is a tuning candidate "
                         "(priors from one-GPU timings of the per-rank shapes at n = 65536, N = 8: 8x1 4.65 ms, 4x2 4.18, 2x4 4.00 per rank)")
    ap.add_argument("--bcast", default=None, choices=["broadcast", "allgather"],
                    help="how a panel of B reaches the ranks: one broadcast, or scatter from rank 0 + all-gather; left unset together "
                         "with --panels 0 the run times three untimed steps of every candidate first and keeps the fastest")
    ap.add_argument("--no-host-path", action="store_true", help="skip the end-to-end legs on host mzd_t (C ABI incl. PCIe)")
    ap.add_argument("--no-elim", action="store_true", help="skip the elimination leg (reduced echelon forms, device resident)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--check", action="store_true", help="spot-check rows of C against the oracle after timing")
    ap.add_argument("--rehearse-one-rank", action="store_true",
                    help="rehearsal on a one-GPU box: run the N > 1 code path (process group, grid, panel tuning, RCCL collectives, gathers, "
                         "self-check, per-rank records) with ONE rank -- the only way the nccl backend can run here, two ranks on one GPU "
                         "are refused by RCCL; the line it prints is not a measurement")
    ap.add_argument("--no-configs", action="store_true",
                    help="N = 1: skip the other single-GPU BASELINE configurations (2, 3, 5) that are timed after the headline loop")
    ap.add_argument("--no-parity", action="store_true", help="skip the sha256 / sampled-row self-checks of the timed products")
    ap.add_argument("--density", default="half", choices=["half", "sparse", "ones"],
                    help="N = 1 only: bit density of the synthetic operands -- half = i.i.d. Bernoulli(1/2) (the metric's "
                         "workload), sparse = 1/64, ones = all ones (clock / data-dependence sanity runs, SURVEY.md section 8d)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks ourselves as fresh child processes (torch.distributed.run, one
        # rank per GPU) BEFORE anything in this process has touched the GPU, relay their output (rank 0 prints the JSON
        # line) and leave with their exit code.  Nothing is exec'ed and this parent never initialises HIP.
        sys.exit(_launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    multi = (int(world) >= 2) or args.rehearse_one_rank  # the distributed code path (one rank: rehearsal only)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the multiply path has no CPU fallback")
    # rehearsal mode: more ranks than GPUs (e.g. 2 ranks on a 1-GPU box with --backend gloo) share device 0
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    if multi:
        if args.rehearse_one_rank:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if multi:
        dist.barrier()
    import m4ri_rust_amd  # noqa: F401
    from m4ri_rust_amd import device
    from m4ri_rust_amd import sharded

    n = args.n
    assert n % (64 * world) == 0
    ldw = n // 64

    def parse_grid(text):
        r_, q_ = (int(x) for x in text.lower().split("x"))
        assert r_ >= 1 and q_ >= 1 and r_ * q_ == world, "--grid %s does not cover %d ranks" % (text, world)
        return r_, q_
    R, Q = parse_grid(args.grid) if (args.grid and multi) else (world, 1)
    rows = n // R
    # all work (kernels and RCCL collectives) is issued under one explicit torch stream: torch orders its
    # collectives against the current stream, and the library launches on the very same hipStream_t
    comp = torch.cuda.Stream()
    torch.cuda.set_stream(comp)
    stream = comp.cuda_stream

    # resident operands (torch owns the memory; the library sees raw device pointers).
    # synthetic data: seeded splitmix64 bits (same stream as the oracle generator); rank r holds rows
    # [r*rows, (r+1)*rows) of the global A (seed 1), B has seed 2.
    def make_a(R_, Q_):
        """Row block i = rank // Q_ of the seeded global A (n / R_ rows)."""
        rows_ = n // R_
        a_t = torch.empty((rows_, ldw), dtype=torch.int64, device="cuda")
        a = device.DMat.from_torch(a_t, n)
        sharded.fill_row_block(a, seed=1, row0=(rank // Q_) * rows_, stream=stream)
        return a_t, a

    grids = {}

    def grid_of(R_, Q_):
        """The process groups of an R_ x Q_ grid (created once, by every rank, in the same order)."""
        if (R_, Q_) not in grids:
            grids[(R_, Q_)] = sharded.Grid(R_, Q_)
        return grids[(R_, Q_)]

    def make_panels(R_, Q_, P_):
        """Buffers of an R_ x Q_ grid with P_ sub-panels per column panel ("tiles" of B, each contiguous, so that one can travel by RCCL
        while the previous one is being multiplied): this rank's sub-panels of ITS column panel of B and of its block of C; on rank 0
        also all of B (b_src[j][p], filled from the seeded generator) and all of C (Cfull[j][p])."""
        rows_, ws_, ncs_ = n // R_, ldw // (Q_ * P_), n // (Q_ * P_)
        b_src_ = None
        if rank == 0:
            b_src_ = [[torch.empty((n, ws_), dtype=torch.int64, device="cuda") for _ in range(P_)] for _ in range(Q_)]
            for j_ in range(Q_):
                for p_ in range(P_):
                    sharded.fill_block(device.DMat.from_torch(b_src_[j_][p_], ncs_), 2, 0, (j_ * P_ + p_) * ws_, n, stream)
        Bp_t_ = b_src_[0] if rank == 0 else [torch.empty((n, ws_), dtype=torch.int64, device="cuda") for _ in range(P_)]
        Cp_t_ = [torch.empty((rows_, ws_), dtype=torch.int64, device="cuda") for _ in range(P_)]
        Bp_ = [device.DMat.from_torch(t, ncs_) for t in Bp_t_]
        Cp_ = [device.DMat.from_torch(t, ncs_) for t in Cp_t_]
        Cfull_t_ = [[torch.empty((n, ws_), dtype=torch.int64, device="cuda") for _ in range(P_)] for _ in range(Q_)] if rank == 0 else None
        return b_src_, Bp_t_, Cp_t_, Bp_, Cp_, Cfull_t_

    def panel_ok(Q_, P_):
        return ldw % (2 * Q_ * P_) == 0 and (n // (Q_ * P_)) % 128 == 0

    A_t, A = make_a(R, Q)

    panel_tuning = None
    if multi and (args.panels <= 0 or args.grid is None) and args.bcast is None:
        # No multi-GPU run of this repository exists yet, so the first one tunes itself: three untimed steps of every candidate
        # (grid of ranks x sub-panels per column panel x how a panel travels), barrier + synchronize around them, the slowest rank's
        # time decides on every rank alike.  Priors from one-GPU timings of the per-rank shapes: DESIGN.md section 6.
        panel_tuning = []
        grid_cands = [(R, Q)] if args.grid else [(world // q_, q_) for q_ in (1, 2, 4, 8) if world % q_ == 0 and n % (64 * (world // q_)) == 0]
        panel_cands = (args.panels,) if args.panels > 0 else (1, 2, 4)
        for (R_, Q_) in grid_cands:
            g_ = grid_of(R_, Q_)
            a_t_, a_ = (A_t, A) if (R_, Q_) == (R, Q) else make_a(R_, Q_)
            for P_ in panel_cands:
                if not panel_ok(Q_, P_) or (Q_ > 1 and Q_ * P_ > 8):
                    continue
                bufs = make_panels(R_, Q_, P_)
                for mode in ("broadcast", "allgather"):
                    if mode == "allgather" and n % R_:
                        continue
                    for it in range(4):  # the first one untimed (arenas, communicators)
                        if it == 1:
                            torch.cuda.synchronize()
                            dist.barrier()
                            t0_ = time.perf_counter()
                        sharded.step_grid(g_, a_, bufs[0], bufs[1], bufs[2], bufs[5], bufs[3], bufs[4], algo=args.algo,
                                          levels=args.levels, stream=stream, bcast=mode)
                    torch.cuda.synchronize()
                    dist.barrier()
                    tm = torch.tensor([(time.perf_counter() - t0_) / 3.0], dtype=torch.float64, device="cuda")
                    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                    panel_tuning.append({"grid": "%dx%d" % (R_, Q_), "panels": P_, "bcast": mode, "ms_per_step": float(tm.item()) * 1e3})
                del bufs
                torch.cuda.empty_cache()
            del a_t_, a_
        bestc = min(panel_tuning, key=lambda c: c["ms_per_step"])
        args.panels, args.bcast = bestc["panels"], bestc["bcast"]
        if parse_grid(bestc["grid"]) != (R, Q):
            R, Q = parse_grid(bestc["grid"])
            rows = n // R
            del A_t, A
            torch.cuda.empty_cache()
            A_t, A = make_a(R, Q)
    if args.bcast is None:
        args.bcast = "broadcast"
    P = (args.panels if args.panels > 0 else (2 if world == 2 else 4)) if multi else 1
    while multi and P > 1 and not panel_ok(Q, P):
        P //= 2
    assert ldw % (2 * Q * P) == 0
    wp, ncp = ldw // (Q * P), n // (Q * P)  # words / columns of one sub-panel
    grid = grid_of(R, Q) if multi else None
    if not multi:
        B_t = torch.empty((n, ldw), dtype=torch.int64, device="cuda")
        C_t = torch.empty((rows, ldw), dtype=torch.int64, device="cuda")
        B = device.DMat.from_torch(B_t, n)
        C = device.DMat.from_torch(C_t, n)
        B.fill_random(2, stream)
        if args.density != "half":
            with torch.cuda.stream(comp):
                if args.density == "ones":
                    A_t.fill_(-1)
                    B_t.fill_(-1)
                else:  # AND of six independent fills: density 2^-6
                    tmp_t = torch.empty_like(A_t)
                    tmp = device.DMat.from_torch(tmp_t, n)
                    for t_, base in ((A_t, 100), (B_t, 200)):
                        for k in range(5):
                            tmp.fill_random(base + k, stream)
                            t_.bitwise_and_(tmp_t)
                    del tmp, tmp_t
    else:
        b_src_t, Bp_t, Cp_t, Bp, Cp, Cfull_t = make_panels(R, Q, P)
    torch.cuda.synchronize()

    step_events = []  # filled during the timed steps only

    def step(record=False):
        if multi:
            ev = [] if record else None
            sharded.step_grid(grid, A, b_src_t, Bp_t, Cp_t, Cfull_t, Bp, Cp, algo=args.algo, levels=args.levels, stream=stream,
                              bcast=args.bcast, events=ev)
            if record:
                step_events.append(ev)
        else:
            device.mul(A, B, C=C, algo=args.algo, param=args.levels, stream=stream)

    # the CPU baseline leg first (N = 1 only): the timed GPU loop then ends the run, so a driver-side utilisation sample
    # taken near the end sees the GPU busy
    cpu_baseline, cpu_sample = (None, None)
    if not multi and not args.no_cpu:
        cpu_baseline, cpu_sample = _cpu_baseline(args)
        if not args.check:
            cpu_sample = None

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    device.prof_enable(True)
    device.prof_read(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record=True)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    device.prof_enable(False)
    launches, kernel_ms = device.prof_read(reset=True)

    dt_local = dt
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    rank_info = None
    if multi:
        mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(),
                "name": torch.cuda.get_device_name(torch.cuda.current_device()), "timed_region_ms": dt_local * 1e3,
                "step_breakdown": sharded.breakdown(step_events)}
        rank_info = [None] * world
        dist.all_gather_object(rank_info, mine)
    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return

    bitops = 2.0 * n * n * n * args.steps
    ms_per_step = dt * 1e3 / args.steps

    sharded_ok = None
    if args.check and multi:
        # rank 0 recomputes every C panel from the full seeded A on its own GPU and compares with what was gathered
        A_full = device.DMat.random(n, n, 1, stream)
        sharded_ok = True
        for j_ in range(Q):
            for pnl in range(P):
                ref = device.mul(A_full, device.DMat.from_torch(b_src_t[j_][pnl], ncp), algo=args.algo, param=args.levels, stream=stream)
                sharded_ok = sharded_ok and device.equal(ref, device.DMat.from_torch(Cfull_t[j_][pnl], ncp), stream)
        del A_full

    self_check = None
    if multi and not args.no_parity:
        self_check = _sharded_self_check(device, sharded, torch, [device.DMat.from_torch(b_src_t[j_][p_], ncp) for j_ in range(Q) for p_ in range(P)],
                                         [Cfull_t[j_][p_] for j_ in range(Q) for p_ in range(P)], n, R, stream)
    parity = None
    if not multi and not args.no_parity and args.density == "half":
        want = _golden_digests().get("sq_%d" % n, {}).get("sha256_c")
        if want:
            parity = _sha256_of(C, stream) == want  # the product of the LAST timed step

    # dominant kernel: the (batched) M4RM tile kernel. Algorithmic bytes of ONE launch = what that
    # launch's products read and write once: batch * (m*l + l*n + m*n)/8 with the leaf dims.
    ncols_launch = n // (Q * P)  # columns of B one launch sees (a sub-panel of the rank's column panel when N > 1)
    levels = sharded.levels_used(rows, n, ncols_launch, args.algo, args.levels)
    # the library may cut the batch of 7^levels leaf products into several launches (chunks that overlap the Strassen passes)
    products = (args.steps * P) or 1
    lps = max(1, int(round(launches / products))) if launches else 1  # tile-kernel launches per product
    mi, li, ni, batch = rows >> levels, n >> levels, ncols_launch >> levels, (7 ** levels) // lps
    alg_bytes_launch = batch * (mi * li + li * ni + mi * ni) / 8.0
    avg_kernel_ms = kernel_ms / max(launches, 1)
    achieved = alg_bytes_launch / (avg_kernel_ms * 1e-3) / 1e9 if launches else 0.0
    # on-chip view: every 8 bits of the inner dimension cost one 16-byte LDS read per 128 columns
    lds_bytes_launch = batch * (mi * (li / 8.0) * (ni / 128.0) * 16.0)
    traffic, traffic_source = None, "not measured in this run (HBM counters need rocprofv3 --pmc passes)"
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tfile):
        try:
            with open(tfile) as f:
                tj = json.load(f)
            if tj.get("n") == n and tj.get("levels") == levels and tj.get("n_gpus") == world:
                traffic = tj.get("hbm_bytes_per_launch") * tj.get("launches_per_product", 1) / lps
                traffic_source = "replayed: profiles/traffic_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this " \
                                 "command in a builder run, %s); not measured in this run" % tj.get("source", "see profiles/INDEX.md")
        except Exception:
            traffic = None
    # the clock the chip holds under the tile kernel: measured with in-kernel stamps in a development build
    # (tools/clock_probe.py), replayed here like `traffic`; the LDS delivers 256 B per clock and CU whatever the clock is
    clock_ghz, clock_source = None, "not measured in this run (needs the development build of the kernel: tools/clock_probe.py)"
    cfile = os.path.join(ROOT, "profiles", "clock_latest.json")
    if os.path.exists(cfile):
        try:
            with open(cfile) as f:
                cj = json.load(f)
            key = {"half": "half", "sparse": "sparse", "ones": "ones"}[args.density]
            if cj.get("n") == n and cj.get("clock_GHz", {}).get(key):
                clock_ghz = float(cj["clock_GHz"][key])
                clock_source = "replayed: profiles/clock_latest.json (%s); not measured in this run" % cj.get("method", "tools/clock_probe.py")
        except Exception:
            clock_ghz = None
    lds_tbps = lds_bytes_launch / (avg_kernel_ms * 1e-3) / 1e12 if launches else 0.0
    out = {
        "metric": "gf2_matmul_bit_ops_per_sec_n%d" % n,
        "value": bitops / dt,
        "unit": "bit-ops/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",  # total work (one n^3 product) is fixed as N grows: rows of A are divided among the ranks
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic" if args.density == "half" else "synthetic, bit density %s" % args.density,
        "config": {
            "workload": "GF(2) %dx%dx%d matmul, inputs resident in HBM, %s%s" % (
                n, n, n, "Strassen(%d levels)-over-M4RM" % levels if levels else "M4RM only",
                "; per step on a %d x %d grid of ranks: B (resident on rank 0) travels in %d column panels of %d sub-panels, rank (i, j) "
                "multiplies its %d-row block i of A by the sub-panels of column panel j, the %d x %d blocks of C are gathered on rank 0"
                % (R, Q, Q, P, rows, rows, ncp) if multi else ""),
            "n": n, "algo": args.algo, "strassen_levels": levels,
            "parallelism": "row-block shard of A over %d GPU(s)%s" % (
                world, " as a %d x %d grid (%d row blocks of A x %d column panels of B; the inner dimension is never split), %d sub-panels "
                       "per column panel: RCCL %s(sub-panel p+1) / gather(C sub-panel p-1) overlap the product of sub-panel p" % (
                    R, Q, R, Q, P, "broadcast" if args.bcast == "broadcast" else "scatter+all_gather")
                if multi else ""),
            "grid": "%dx%d" % (R, Q) if multi else None,
        },
        "roofline": {
            "bound": "hbm",
            # v8 (4096 x 512 tiles) on row-group-packed A for tall leaves; the launcher picks tile height / variant by shape
            "kernel": "M4RM tile kernel gf2_m4rm_kernel_v8 (batch of %d leaf products %dx%dx%d; the products of the last, incomplete "
                      "round of 256 tiles run in a second launch with shorter tiles cut into stream-K segments)" % (batch, mi, li, ni),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_of_measured_copy": achieved / HBM_COPY_GBS,
            "traffic": traffic, "traffic_source": traffic_source,
            "launches": launches, "avg_launch_ms": avg_kernel_ms,
            "algorithmic_bytes_per_launch": alg_bytes_launch,
            "note": "this kernel is LDS-bound, not HBM-bound (see onchip); HBM fraction is small by construction",
            "onchip": {
                "bound": "lds", "achieved": lds_tbps,
                "peak": LDS_PEAK_TBS, "unit": "TB/s",
                "frac": lds_tbps / LDS_PEAK_TBS,
                # the same bytes against what the LDS can deliver at the clock the chip actually holds under this kernel
                # (256 B per clock and CU x 256 CUs): random operands run at ~2.2 GHz, sparse / constant ones at ~2.37
                "clock_GHz": clock_ghz,
                "peak_at_measured_clock": (LDS_BYTES_PER_CLK * clock_ghz * 1e9 / 1e12) if clock_ghz else None,
                "frac_at_measured_clock": (lds_tbps / (LDS_BYTES_PER_CLK * clock_ghz * 1e9 / 1e12)) if clock_ghz else None,
                "clock_source": clock_source,
            },
        },
        "hbm_equiv_GBps_whole_step": 3.0 * n * n / 8.0 / (ms_per_step * 1e-3) / 1e9,
    }
    if sharded_ok is not None:
        out["sharded_result_matches_single_gpu"] = bool(sharded_ok)
    if parity is not None:
        out["parity_sha256_ok"] = bool(parity)
        out["parity_note"] = "sha256 of the last timed %d^3 product == tests/golden/digests_large.json[sq_%d]" % (n, n)
    if multi:
        out["ranks_seen"] = dist.get_world_size()
        out["backend"] = dist.get_backend()
        out["ranks"] = rank_info
        # where a step's time goes, per-rank maxima of the marks on the compute stream (per column panel: waiting for B, the local
        # product; then the wait for the gathers of C): the longer leg is what to tune next
        bds = [r["step_breakdown"] for r in rank_info if r and r.get("step_breakdown")]
        if bds:
            Pn = len(bds[0]["wait_b_ms"])
            out["step_breakdown"] = {
                "wait_b_ms": [max(b["wait_b_ms"][p] for b in bds) for p in range(Pn)],
                "product_ms": [max(b["product_ms"][p] for b in bds) for p in range(Pn)],
                "gather_tail_ms": max(b["gather_tail_ms"] for b in bds),
                "step_ms": max(b["step_ms"] for b in bds),
                "note": "per-rank maxima; marks recorded on the compute stream inside the timed steps"}
        if panel_tuning is not None:
            out["panel_tuning"] = {"candidates": panel_tuning, "chosen": {"grid": "%dx%d" % (R, Q), "panels": P, "bcast": args.bcast},
                                   "note": "three untimed steps per candidate before the timed loop; --grid / --panels / --bcast pin a choice"}
        if self_check is not None:
            out["self_check"] = self_check
            out["parity_rows_ok"] = self_check["ok"]
    if not multi and levels > 0 and launches:
        # the rest of a step is the Strassen split / merge passes (same stream, serial): HBM-streaming kernels whose
        # bytes are known exactly (every pass reads its sources once and writes its destinations once; the library's own count)
        pass_bytes = float(device._lib.lib().gf2_strassen_pass_bytes(n, n, n, levels))
        pass_ms = max(ms_per_step - kernel_ms / args.steps, 1e-9)
        out["strassen_passes"] = {"bound": "hbm", "ms_per_step": pass_ms, "bytes_per_step": pass_bytes,
                                  "achieved": pass_bytes / (pass_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": pass_bytes / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "note": "step time minus the tile-kernel launch; gf2_strassen_split3 / merge3 (three fused levels, a "
                                          "virtual fourth on top) and the single-level merge"}

    if cpu_baseline is not None:
        out["cpu_baseline"] = cpu_baseline
        if args.check and cpu_sample is not None and args.cpu_n <= n:
            # the GPU must reproduce the CPU sample product bit for bit
            import numpy as np
            cn = args.cpu_n
            Ad, Bd = device.DMat.random(cn, cn, 1), device.DMat.random(cn, cn, 2)
            out["cpu_baseline"]["gpu_matches_cpu_sample"] = bool(
                np.array_equal(device.mul(Ad, Bd, algo=args.algo, param=args.levels).to_words(), cpu_sample))
    if not multi and not args.no_configs and args.density == "half":
        out["configs"] = _extra_configs(device, torch, stream)
    if not multi and not args.no_elim and args.density == "half":
        out["elimination"] = _elimination(device, torch)
    if not multi and not args.no_host_path and args.density == "half":
        del A, B, C, A_t, B_t, C_t  # the resident operands and (gf2_trim) the 15 GiB arena go back first
        torch.cuda.empty_cache()
        device._lib.lib().gf2_trim()
        # (after such a hipFree the copies of the host path run a third slower for a while -- mzd_transpose 65536^2 16.1 ms against
        # 12.3 in a process that has freed nothing, or that waits first: tools/ht_probe.py, profiles/r04_host_path.txt -- a property
        # of freeing, not of the path: _host_path waits until its PCIe rate measurement has settled)
        time.sleep(1.0)
        out["host_path"] = _host_path(torch)
    print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
