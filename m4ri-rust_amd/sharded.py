"""Multi-GPU host layer: C = A*B with A and C sharded by row blocks over the ranks of one node.

C[R,:] = A[R,:] * B for disjoint row blocks R, so the units are independent given B: the only
exchange is broadcast(B) from rank 0 and gather(C blocks) onto rank 0 (north_star: "shard
row-blocks of A across the 8 GPUs of one node with RCCL broadcast of B tiles and gather of C over
xGMI").  One process per GPU, torch.distributed ("nccl" == RCCL on ROCm); no XOR all-reduce is
needed because the inner dimension is never split.

The collective choreography is independent of where the local product runs, so it is testable on
CPU with the gloo backend by injecting `local_mul` (tests/test_sharded.py injects the oracle);
the default `local_mul` is the HIP path and fails loudly without a GPU.
"""
import ctypes

from . import _lib, device


def fill_row_block(A, seed, row0, stream=None):
    """Rows [row0, row0 + A.nrows) of the seeded global matrix (same stream as the oracle generator)."""
    _lib.check(_lib.lib().gf2_dmat_fill_random_rows(A._on(stream), seed, row0, stream), "gf2_dmat_fill_random_rows")


def levels_used(m, l, n, algo, levels):
    return _lib.lib().gf2_strassen_levels(m, l, n, device.ALGOS[algo], levels)


def _hip_local_mul(a_block, b, c_block, ncols_inner, ncols_out, algo="auto", levels=0):
    """Default local product: torch int64 CUDA tensors -> gf2_mul_dev on torch's current stream (the stream
    torch orders its collectives against; if that is the default stream the handle is NULL = legacy default
    stream, which synchronises with every other stream)."""
    import torch
    device.require_gpu()
    stream = torch.cuda.current_stream().cuda_stream
    A = device.DMat.from_torch(a_block, ncols_inner)
    B = device.DMat.from_torch(b, ncols_out)
    C = device.DMat.from_torch(c_block, ncols_out)
    device.mul(A, B, C=C, algo=algo, param=levels, stream=stream)


def mul_row_sharded(a_block, b, c_block, c_full, ncols_inner, ncols_out, local_mul=None, group=None, **kw):
    """One sharded product.

    a_block : (rows_local, ceil(l/64)) int64 tensor, this rank's row block of A
    b       : (l, ceil(n/64)) int64 tensor; contents significant on rank 0, overwritten elsewhere
    c_block : (rows_local, ceil(n/64)) int64 tensor, receives this rank's block of C
    c_full  : on rank 0 a (rows_total, ceil(n/64)) tensor receiving all of C, None elsewhere
    """
    import torch.distributed as dist
    local_mul = local_mul or _hip_local_mul
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world > 1:
        dist.broadcast(b, src=0, group=group)
    local_mul(a_block, b, c_block, ncols_inner, ncols_out, **kw)
    if world > 1:
        if rank == 0:
            dist.gather(c_block, gather_list=list(c_full.chunk(world, dim=0)), dst=0, group=group)
        else:
            dist.gather(c_block, gather_list=None, dst=0, group=group)
    elif c_full is not None:
        c_full.copy_(c_block)


def fill_block(M, seed, row0, col_word0, full_ncols, stream=None):
    """Rows [row0, ..) x 64-bit words [col_word0, ..) of the seeded global matrix with `full_ncols` columns."""
    _lib.check(_lib.lib().gf2_dmat_fill_random_block(M._on(stream), seed, row0, col_word0, full_ncols, stream),
               "gf2_dmat_fill_random_block")


class _Chain:
    """Several asynchronous collectives that complete in order: wait() on the last one is enough on GPUs (one
    communication stream), gloo wants every handle waited."""

    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


def distribute_panel(b_panel, mode="broadcast", group=None):
    """Start moving one column panel of B from rank 0 to every rank; returns a handle with .wait().

    "broadcast": one dist.broadcast (a pipeline along RCCL's rings).
    "allgather": rank 0 scatters row shards of the panel (its links to all peers in parallel), then every rank
    all-gathers the shards (uses all seven xGMI links of every GPU instead of one chain).
    Both leave the same bytes in b_panel on every rank."""
    import torch.distributed as dist
    if mode == "broadcast":
        return dist.broadcast(b_panel, src=0, group=group, async_op=True)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    shards = list(b_panel.chunk(world, dim=0))
    assert len(shards) == world and all(s.shape == shards[0].shape for s in shards), "panel rows must divide by the world size"
    mine = shards[rank]
    w1 = dist.scatter(mine, scatter_list=shards if rank == 0 else None, src=0, group=group, async_op=True)
    src = mine
    if dist.get_backend(group) != "nccl":  # gloo runs asynchronous operations in no particular order, and dislikes aliased buffers
        w1.wait()
        src = mine.clone()
    w2 = dist.all_gather(shards, src, group=group, async_op=True)
    return _Chain([w1, w2])


def mul_row_sharded_pipelined(a_block, b_panels, c_panels, c_full_panels, ncols_inner, ncols_panel, local_mul=None,
                              group=None, bcast="broadcast", **kw):
    """Row-sharded product with B moved in column panels ("tiles"): panel p+1 is on the wire while panel p is being
    multiplied, and the gather of C panel p overlaps the product of panel p+1.

    b_panels[p]      : (l, ceil(ncols_panel/64)) int64, contiguous; columns [p*ncols_panel, (p+1)*ncols_panel) of B
                       (significant on rank 0, overwritten elsewhere)
    c_panels[p]      : (rows_local, ceil(ncols_panel/64)) int64, this rank's rows of C panel p
    c_full_panels[p] : on rank 0 (rows_total, ceil(ncols_panel/64)) receiving C panel p, None elsewhere
    """
    import torch.distributed as dist
    local_mul = local_mul or _hip_local_mul
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    P = len(b_panels)
    if world == 1:
        for p in range(P):
            local_mul(a_block, b_panels[p], c_panels[p], ncols_inner, ncols_panel, **kw)
            if c_full_panels is not None:
                c_full_panels[p].copy_(c_panels[p])
        return
    bcasts = [distribute_panel(b_panels[p], bcast, group) for p in range(P)]
    gathers = []
    for p in range(P):
        bcasts[p].wait()  # orders the consumer (current stream on GPUs) behind the arrival of panel p
        local_mul(a_block, b_panels[p], c_panels[p], ncols_inner, ncols_panel, **kw)
        glist = list(c_full_panels[p].chunk(world, dim=0)) if rank == 0 else None
        gathers.append(dist.gather(c_panels[p], gather_list=glist, dst=0, group=group, async_op=True))
    for w in gathers:
        w.wait()


def step_pipelined(A, Bp_t, Cp_t, Cfull_t, Bp, Cp, algo="auto", levels=0, stream=None, bcast="broadcast", events=None):
    """bench.py's timed step for N > 1 ranks, B in column panels (DMat wrappers pre-built).

    events: None, or a list that receives this step's torch.cuda.Event marks on the compute stream --
    [start, (B panel p arrived, product p enqueued) for every p, all gathers done] -- from which `breakdown()` derives, per panel,
    how long the compute stream waited for B and how long the product took, and the tail spent waiting for the gathers: the
    first real multi-GPU run then says by itself which leg (RCCL transfer or local product) bounds a step."""
    import torch
    import torch.distributed as dist
    world, rank, P = dist.get_world_size(), dist.get_rank(), len(Bp_t)

    def mark():
        if events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()  # torch's current stream = the bench's compute stream
            events.append(e)
    mark()
    bcasts = [distribute_panel(Bp_t[p], bcast) for p in range(P)]
    gathers = []
    for p in range(P):
        bcasts[p].wait()
        mark()
        device.mul(A, Bp[p], C=Cp[p], algo=algo, param=levels, stream=stream)
        mark()
        glist = list(Cfull_t[p].chunk(world, dim=0)) if rank == 0 else None
        gathers.append(dist.gather(Cp_t[p], gather_list=glist, dst=0, async_op=True))
    for w in gathers:
        w.wait()
    mark()


def breakdown(step_events):
    """Per-step means (ms) out of the marks `step_pipelined` recorded: {"wait_b_ms": [per panel], "product_ms": [per panel],
    "gather_tail_ms": x, "step_ms": y}.  Call after torch.cuda.synchronize()."""
    if not step_events:
        return None
    P = (len(step_events[0]) - 2) // 2
    wait_b, prod, tail, total = [0.0] * P, [0.0] * P, 0.0, 0.0
    for ev in step_events:
        for p in range(P):
            wait_b[p] += ev[2 * p].elapsed_time(ev[2 * p + 1])
            prod[p] += ev[2 * p + 1].elapsed_time(ev[2 * p + 2])
        tail += ev[2 * P].elapsed_time(ev[2 * P + 1])
        total += ev[0].elapsed_time(ev[2 * P + 1])
    k = float(len(step_events))
    return {"wait_b_ms": [x / k for x in wait_b], "product_ms": [x / k for x in prod], "gather_tail_ms": tail / k, "step_ms": total / k}


def step(A_t, B_t, C_t, Cfull_t, A, B, C, algo="auto", levels=0, stream=None):
    """bench.py's timed step for N > 1 ranks (DMat wrappers are pre-built to keep Python overhead out)."""
    import torch.distributed as dist
    world = dist.get_world_size()
    dist.broadcast(B_t, src=0)
    device.mul(A, B, C=C, algo=algo, param=levels, stream=stream)
    if dist.get_rank() == 0:
        dist.gather(C_t, gather_list=list(Cfull_t.chunk(world, dim=0)), dst=0)
    else:
        dist.gather(C_t, gather_list=None, dst=0)
