"""Multi-GPU choreography on CPU: world size 2, gloo backend.  The row-block sharding, broadcast(B) and
gather(C) of m4ri-rust_amd/sharded.py run exactly as on GPUs; only the local product is injected (the CPU
oracle stands in for the HIP kernel here -- tests may do that, the product may not)."""
import os
import socket
import sys

import numpy as np
import pytest

import gf2util as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, m, l, n, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import gf2util as gg
    import m4ri_rust_amd  # noqa: F401
    from m4ri_rust_amd import sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = m // world
    a_full = gg.random_words(m, l, 1)
    a_block = torch.from_numpy(a_full[rank * rows:(rank + 1) * rows].copy().view(np.int64))
    b = torch.from_numpy(gg.random_words(l, n, 2).view(np.int64).copy()) if rank == 0 else torch.zeros(
        (l, gg.width(n)), dtype=torch.int64)
    c_block = torch.zeros((rows, gg.width(n)), dtype=torch.int64)
    c_full = torch.zeros((m, gg.width(n)), dtype=torch.int64) if rank == 0 else None

    def oracle_local_mul(a_t, b_t, c_t, ncols_inner, ncols_out, **_kw):
        a = np.ascontiguousarray(a_t.numpy().view(np.uint64))
        bb = np.ascontiguousarray(b_t.numpy().view(np.uint64))
        c = gg.o_mul_m4rm(a, bb, a.shape[0], ncols_inner, ncols_out)
        c_t.copy_(torch.from_numpy(c.view(np.int64)))

    sharded.mul_row_sharded(a_block, b, c_block, c_full, l, n, local_mul=oracle_local_mul)
    if rank == 0:
        np.save(out_path, c_full.numpy().view(np.uint64))
    dist.barrier()
    dist.destroy_process_group()


def _worker_panels(rank, world, port, m, l, n, P, out_path, bcast="broadcast"):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import gf2util as gg
    import m4ri_rust_amd  # noqa: F401
    from m4ri_rust_amd import sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, wp, ncp = m // world, gg.width(n) // P, n // P
    a_block = torch.from_numpy(gg.random_words(m, l, 1)[rank * rows:(rank + 1) * rows].copy().view(np.int64))
    b_full = gg.random_words(l, n, 2)
    b_panels = [torch.from_numpy(b_full[:, p * wp:(p + 1) * wp].copy().view(np.int64)) if rank == 0
                else torch.zeros((l, wp), dtype=torch.int64) for p in range(P)]
    c_panels = [torch.zeros((rows, wp), dtype=torch.int64) for _ in range(P)]
    c_full = [torch.zeros((m, wp), dtype=torch.int64) for _ in range(P)] if rank == 0 else None

    def oracle_local_mul(a_t, b_t, c_t, ncols_inner, ncols_out, **_kw):
        a = np.ascontiguousarray(a_t.numpy().view(np.uint64))
        bb = np.ascontiguousarray(b_t.numpy().view(np.uint64))
        c_t.copy_(torch.from_numpy(gg.o_mul_m4rm(a, bb, a.shape[0], ncols_inner, ncols_out).view(np.int64)))

    for _ in range(2):  # two steps back to back: panels are re-broadcast into the same buffers
        sharded.mul_row_sharded_pipelined(a_block, b_panels, c_panels, c_full, l, ncp, local_mul=oracle_local_mul, bcast=bcast)
    if rank == 0:
        np.save(out_path, np.concatenate([t.numpy().view(np.uint64) for t in c_full], axis=1))
    dist.barrier()
    dist.destroy_process_group()


def _worker_grid(rank, world, port, m, l, n, R, Q, P, out_path, bcast):
    """One rank of an R x Q grid (sharded.mul_grid_pipelined): row block i = rank // Q of A, column panel j = rank % Q of B."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import gf2util as gg
    import m4ri_rust_amd  # noqa: F401
    from m4ri_rust_amd import sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grid = sharded.Grid(R, Q)
    i = rank // Q
    rows, ws, ncs = m // R, gg.width(n) // (Q * P), n // (Q * P)
    a_block = torch.from_numpy(gg.random_words(m, l, 1)[i * rows:(i + 1) * rows].copy().view(np.int64))
    b_full = gg.random_words(l, n, 2)
    b_src = None
    if rank == 0:
        b_src = [[torch.from_numpy(b_full[:, (j * P + p) * ws:(j * P + p + 1) * ws].copy().view(np.int64)) for p in range(P)] for j in range(Q)]
    b_panels = b_src[0] if rank == 0 else [torch.zeros((l, ws), dtype=torch.int64) for _ in range(P)]
    c_panels = [torch.zeros((rows, ws), dtype=torch.int64) for _ in range(P)]
    c_full = [[torch.zeros((m, ws), dtype=torch.int64) for _ in range(P)] for _ in range(Q)] if rank == 0 else None

    def oracle_local_mul(a_t, b_t, c_t, ncols_inner, ncols_out, **_kw):
        a = np.ascontiguousarray(a_t.numpy().view(np.uint64))
        bb = np.ascontiguousarray(b_t.numpy().view(np.uint64))
        c_t.copy_(torch.from_numpy(gg.o_mul_m4rm(a, bb, a.shape[0], ncols_inner, ncols_out).view(np.int64)))

    for _ in range(2):  # two steps back to back: the panels travel into the same buffers again
        sharded.mul_grid_pipelined(grid, a_block, b_src, b_panels, c_panels, c_full, l, ncs, local_mul=oracle_local_mul, bcast=bcast)
    if rank == 0:
        np.save(out_path, np.concatenate([c_full[j][p].numpy().view(np.uint64) for j in range(Q) for p in range(P)], axis=1))
        # rank 0's copy of B must have survived the all-gathers into its own panel buffers
        assert all(np.array_equal(b_src[j][p].numpy().view(np.uint64), b_full[:, (j * P + p) * ws:(j * P + p + 1) * ws])
                   for j in range(Q) for p in range(P))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bcast", ["broadcast", "allgather"])
@pytest.mark.parametrize("R,Q,P", [(4, 1, 1), (2, 2, 1), (2, 2, 2), (1, 4, 1), (4, 1, 2)], ids=lambda v: str(v))
def test_grid_world4_gloo(tmp_path, built, R, Q, P, bcast):
    """R x Q grids of four ranks (round 5): rank (i, j) multiplies row block i of A with column panel j of B only; 4 x 1 (the
    row-block scheme), 2 x 2 and 1 x 4 must all deliver the same C on rank 0 -- the oracle's -- for both ways B travels."""
    import torch.multiprocessing as mp
    m, l, n = 256, 192, 512
    out = str(tmp_path / "c.npy")
    mp.spawn(_worker_grid, args=(4, _free_port(), m, l, n, R, Q, P, out, bcast), nprocs=4, join=True)
    ref = g.o_mul_m4rm(g.random_words(m, l, 1), g.random_words(l, n, 2), m, l, n)
    assert np.array_equal(np.load(out), ref)


@pytest.mark.parametrize("R,Q", [(2, 1), (1, 2)], ids=["2x1", "1x2"])
def test_grid_world2_gloo(tmp_path, built, R, Q):
    import torch.multiprocessing as mp
    m, l, n = 128, 200, 256
    out = str(tmp_path / "c.npy")
    mp.spawn(_worker_grid, args=(2, _free_port(), m, l, n, R, Q, 2, out, "allgather"), nprocs=2, join=True)
    ref = g.o_mul_m4rm(g.random_words(m, l, 1), g.random_words(l, n, 2), m, l, n)
    assert np.array_equal(np.load(out), ref)


@pytest.mark.parametrize("bcast", ["broadcast", "allgather"])
@pytest.mark.parametrize("P,n", [(2, 256), (4, 512)], ids=["2panels", "4panels"])
def test_pipelined_column_panels_world2_gloo(tmp_path, built, P, n, bcast):
    """B moved in column panels with asynchronous broadcast / gather (the N > 1 step of bench.py: 2 panels on 2 GPUs,
    4 above)."""
    import torch.multiprocessing as mp
    m, l = 128, 192
    out = str(tmp_path / "c.npy")
    mp.spawn(_worker_panels, args=(2, _free_port(), m, l, n, P, out, bcast), nprocs=2, join=True)
    ref = g.o_mul_m4rm(g.random_words(m, l, 1), g.random_words(l, n, 2), m, l, n)
    assert np.array_equal(np.load(out), ref)


@pytest.mark.parametrize("shape", [(128, 200, 130), (64, 64, 64)], ids=lambda s: "x".join(map(str, s)))
def test_row_sharded_product_world2_gloo(tmp_path, shape, built):
    import torch.multiprocessing as mp
    m, l, n = shape
    out = str(tmp_path / "c.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, m, l, n, out), nprocs=2, join=True)
    c = np.load(out)
    ref = g.o_mul_m4rm(g.random_words(m, l, 1), g.random_words(l, n, 2), m, l, n)
    assert np.array_equal(c, ref)


def test_default_local_mul_fails_loudly_without_gpu(built):
    import torch
    import m4ri_rust_amd  # noqa: F401
    from m4ri_rust_amd import device, sharded
    if device.device_count() > 0:
        pytest.skip("GPU present")
    a = torch.zeros((4, 1), dtype=torch.int64)
    with pytest.raises(m4ri_rust_amd._lib.HipError):
        sharded._hip_local_mul(a, a, a, 4, 4)


def test_row_block_generator_matches_full_matrix():
    """fill_row_block(row0) must reproduce rows [row0, row0+rows) of the seeded global matrix (oracle stream)."""
    full = g.random_words(96, 130, 5)
    w = g.width(130)
    t = np.arange(32 * w, dtype=np.uint64) + np.uint64(32 * w)  # rows 32..63
    blk = g.splitmix64(5, t).reshape(32, w)
    blk[:, -1] &= np.uint64((1 << (130 % 64)) - 1)
    assert np.array_equal(blk, full[32:64])


# ---- the driver's own invocation, rehearsed on the GPU box ---------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("bcast,grid", [("broadcast", None), ("allgather", None), (None, None), ("broadcast", "1x2"), ("allgather", "1x2"),
                                        ("allgather", "2x1")])
def test_bench_two_ranks_from_a_bare_shell(bcast, grid):
    """`python bench.py --gpus 2 ...` with WORLD_SIZE unset -- exactly how the driver starts the scaling runs -- must launch
    its own two ranks (torch.distributed.run children; the parent never touches the GPU), run the sharded step (RCCL / gloo
    movement of B's panels, HIP product of every rank's block, gather of C) and print ONE JSON line whose sharded result equals
    the single-GPU product.  On the one-GPU test box the two ranks share device 0 and talk over gloo; on a multi-GPU node
    the same command with --backend nccl is the real thing.  Grids (round 5): 2x1 = two row blocks of A against all of B (the
    row-block scheme), 1x2 = all of A against two column panels of B; unset, both are tuning candidates."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--check", "--dim", "8192",
           "--no-cpu", "--steps", "2", "--warmup", "1"] + (["--bcast", bcast] if bcast else []) + (["--grid", grid] if grid else [])
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["sharded_result_matches_single_gpu"] is True
    assert out["self_check"]["ok"] is True and out["parity_rows_ok"] is True
    assert out["scaling"] == "strong" and out["value"] > 0 and out["config"]["n"] == 8192
    assert out["config"]["grid"] == (grid or "2x1") or bcast is None
    # the line says where a step's time went (marks on the compute stream, per-rank maxima) ...
    bd = out["step_breakdown"]
    assert len(bd["wait_b_ms"]) == len(bd["product_ms"]) >= 1 and bd["step_ms"] > 0 and bd["gather_tail_ms"] >= 0
    assert all(r["step_breakdown"] for r in out["ranks"])
    if bcast is None:  # ... and, with --grid / --panels / --bcast left alone, the run chose its plan from its own timings
        tune = out["panel_tuning"]
        assert {(c["grid"], c["panels"], c["bcast"]) for c in tune["candidates"]} == {(gr, p, b) for gr in ("2x1", "1x2") for p in (1, 2, 4)
                                                                                      for b in ("broadcast", "allgather")}
        best = min(tune["candidates"], key=lambda c: c["ms_per_step"])
        assert tune["chosen"] == {"grid": best["grid"], "panels": best["panels"], "bcast": best["bcast"]}
        assert out["config"]["grid"] == best["grid"]
    else:
        assert "panel_tuning" not in out


@pytest.mark.gpu
def test_rccl_one_rank_collectives():
    """The collectives of the multi-GPU step through RCCL ITSELF (backend nccl), with the one rank a one-GPU box can hold (round 5;
    until then no RCCL call had run for this repository): asynchronous broadcast, scatter + all-gather and gather on the compute
    stream, the library's products between them on the same stream, source and destination buffers aliased as on rank 0 of a
    real run; C must equal the product of the whole operands for 1, 2 and 4 sub-panels and both ways B travels
    (tools/rccl_one_rank.py; reference caller binary_matrix.rs:459-472 -> strassen.rs:18)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_one_rank.py")], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "rccl_one_rank: ok" in r.stdout and "MISMATCH" not in r.stdout
    assert r.stdout.count("C matches") == 12 and "via nccl" in r.stdout


@pytest.mark.gpu
def test_bench_distributed_path_on_rccl_with_one_rank():
    """`bench.py --rehearse-one-rank`: the WHOLE N > 1 code path of the bench -- process group on the nccl backend (RCCL), grid, the
    self-tuning over sub-panels and transports, the timed sharded steps, the MAX reduction of the timed region, per-rank records by
    all_gather_object, sampled-row self-check, comparison with the single-GPU product -- with the one rank a one-GPU box can give RCCL
    (two ranks on one GPU are refused: 'Duplicate GPU detected').  The two-rank rehearsals above run the same code over gloo."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MASTER_PORT"] = str(_free_port())
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-one-rank", "--check", "--dim", "8192", "--no-cpu", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["backend"] == "nccl" and out["ranks_seen"] == 1 and out["config"]["grid"] == "1x1"
    assert out["sharded_result_matches_single_gpu"] is True and out["self_check"]["ok"] is True and out["parity_rows_ok"] is True
    tune = out["panel_tuning"]
    assert {(c["panels"], c["bcast"]) for c in tune["candidates"]} == {(p, b) for p in (1, 2, 4) for b in ("broadcast", "allgather")}
    assert out["step_breakdown"]["step_ms"] > 0 and len(out["ranks"]) == 1
