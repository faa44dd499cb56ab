"""The launch planner's time model against the clock: `pick_levels` / `plan_tiles` choose Strassen level
counts and tile variants by MODELLED time (m4ri-rust_amd/csrc/m4ri_hip_api.cpp: level_time_model), with constants fitted on
measurements.  In round 3 the single-level plan ran 2.1 x above its model (unpacked A leaves) and nothing noticed; this test
holds every (shape, level count) of a small sweep to the model, so that a plan the planner is asked to price cannot drift that
far again.  Reference entry points behind it: mzd_mul (m4ri-sys/src/strassen.rs:8-18), mzd_mul_m4rm (brilliantrussian.rs:210-216)."""
import time

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(built):
    import m4ri_rust_amd  # noqa: F401
    from m4ri_rust_amd import device
    device.require_gpu()
    return device

SHAPES = [(8192, 8192, 8192), (12288, 12288, 12288), (16384, 16384, 16384), (24576, 24576, 24576), (8192, 32768, 16384)]
# The model may be at most 20 % optimistic (the round-3 failure: a plan that ran 2.1 x above its price) and at most 35 % pessimistic:
# its constants were fitted at the clock the chip holds under a long load, products of a few hundred microseconds run before the
# clock has come down, and the boxes of the pool differ on exactly those (12288^3 without levels: 0.291 ms on one box, 0.362 on
# another, modelled 0.371)
OPTIMISTIC, PESSIMISTIC = 0.20, 0.35


@pytest.mark.parametrize("m,l,n", SHAPES)
def test_model_time_tracks_measured_time(dev, m, l, n):
    import torch
    from m4ri_rust_amd import sharded
    lib = dev._lib.lib()
    A, B, C = dev.DMat.random(m, l, 1), dev.DMat.random(l, n, 2), dev.DMat(m, n)
    for _ in range(40):  # clocks up (an idle GPU runs the first milliseconds slow)
        dev.mul(A, B, C=C, algo="m4rm")
    torch.cuda.synchronize()
    report = []
    for L in range(0, 4):
        algo = "m4rm" if L == 0 else "strassen"
        if L and sharded.levels_used(m, l, n, "strassen", L) != L:
            break
        model = lib.gf2_model_time(m, l, n, L)
        assert model > 0
        best = None
        for _ in range(3):
            dev.mul(A, B, C=C, algo=algo, param=L)
            torch.cuda.synchronize()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                dev.mul(A, B, C=C, algo=algo, param=L)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            best = dt if best is None else min(best, dt)
        report.append((L, best, model))
    bad = [(L, round(t * 1e3, 3), round(mod * 1e3, 3)) for L, t, mod in report if mod < (1 - OPTIMISTIC) * t or mod > (1 + PESSIMISTIC) * t]
    assert not bad, "levels whose modelled time is more than 20 %% below / 35 %% above the measured one (L, measured ms, model ms): %s" % bad
    # and the automatic choice is within 5 % of the best explicit level count
    auto = lib.gf2_strassen_levels(m, l, n, dev.ALGOS["auto"], 0)
    t_auto = next(t for L, t, _ in report if L == auto) if any(L == auto for L, _, _ in report) else None
    if t_auto is not None:
        assert t_auto <= 1.05 * min(t for _, t, _ in report), (auto, [(L, round(t * 1e3, 3)) for L, t, _ in report])


def test_host_schedule_model_tracks_measured_time(built, monkeypatch):
    """The schedules of a large product on HOST matrices (row blocks; slabs of the inner dimension; two row groups through the slabs;
    round 5) are chosen by playing them through with the planner's time model and an assumed PCIe rate (plan_host_product).  At
    32768^3 every schedule's modelled end must be within 20 % of the clock, and the schedule the model picks must not lose more than
    5 % to the best one measured (measured margins on the round's boxes: model / clock 0.93 ... 1.01, the pick 0-1 % behind the best; the
    model's link rate is a constant, the boxes' PCIe rates differ by a few per cent).  Reference entry point: mzd_mul on host mzd_t (m4ri-sys/src/strassen.rs:18 <- binary_matrix.rs:459-472)."""
    import ctypes
    import m4ri_rust_amd as pkg
    from m4ri_rust_amd import device
    device.require_gpu()
    L = pkg._lib.lib()
    n = 32768
    te = (ctypes.c_double * 12)()
    monkeypatch.setenv("M4RI_HIP_HOST_PLAN", "0")
    chosen = L.gf2_host_plan_model(n, n, n, 0, 0, te)
    assert 1 <= chosen <= 12 and all(t > 0 for t in te)
    A, B = pkg.BinMatrix.random(n, n), pkg.BinMatrix.random(n, n)
    C = pkg.BinMatrix.zero(n, n)
    measured = {}
    for plan in range(1, 13):
        monkeypatch.setenv("M4RI_HIP_HOST_PLAN", str(plan))
        L.mzd_mul(C.mzd, A.mzd, B.mzd, 0)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            assert L.mzd_mul(C.mzd, A.mzd, B.mzd, 0)
            ts.append(time.perf_counter() - t0)
        measured[plan] = min(ts)
    report = [(p, round(measured[p] * 1e3, 2), round(te[p - 1] * 1e3, 2)) for p in measured]
    bad = [r for r in report if not 0.80 * r[1] <= r[2] <= 1.20 * r[1]]
    assert not bad, "schedules whose modelled end is more than 20 %% off the measured one (plan, measured ms, model ms): %s of %s" % (bad, report)
    assert measured[chosen] <= 1.05 * min(measured.values()), (chosen, report)
