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


# ---------------------------------------------------------------------------------------------
# Two-dimensional grids of ranks (round 5): R row blocks of A x Q column panels of B, R * Q = world.
#
# Rank r = i * Q + j keeps row block i of A resident (rows / R rows), receives ONLY column panel j of B (cols / Q columns, in P
# sub-panels for the overlap of transfer and product), multiplies once per sub-panel, and the blocks of C are gathered on rank 0.
# Still north_star's decomposition -- rows of A sharded, tiles of B by RCCL, gather of C, the inner dimension never split, no
# reduction -- but a rank's product is squarer (one-GPU timings at n = 65536, profiles/r04_size_sweep.txt: 8192 x 65536 x 65536
# 4.65 ms, 16384 x 65536 x 32768 4.18, 32768 x 65536 x 16384 4.00) and it needs 1 / Q of B: 128 MiB instead of 512 per rank and step
# on a 2 x 4 grid.  Q = 1 is the row-block scheme above.
# ---------------------------------------------------------------------------------------------

class Grid:
    """The process groups of an R x Q grid.  Collective: EVERY rank of the world constructs it with the same (R, Q), in the same order
    relative to other group creations."""

    def __init__(self, R, Q, world=None, rank=None):
        import torch.distributed as dist
        self.world = dist.get_world_size() if world is None else world
        self.rank = dist.get_rank() if rank is None else rank
        assert R >= 1 and Q >= 1 and R * Q == self.world, "grid %d x %d does not cover %d ranks" % (R, Q, self.world)
        self.R, self.Q = R, Q
        self.i, self.j = self.rank // Q, self.rank % Q
        # column group j: the R ranks that share column panel j of B; bgroup j: rank 0 (where B lives) + column group j
        self.col_ranks = [[i * Q + j for i in range(R)] for j in range(Q)]
        self.col_groups, self.bgroups = [], []
        for j in range(Q):
            if Q == 1:
                self.col_groups.append(None)  # the default group: exactly the row-block scheme
                self.bgroups.append(None)
                continue
            cg = dist.new_group(ranks=self.col_ranks[j])
            self.col_groups.append(cg)
            self.bgroups.append(cg if j == 0 else dist.new_group(ranks=[0] + self.col_ranks[j]))
        self._side = None

    def block_of(self, r):
        return r // self.Q, r % self.Q

    def side_stream(self, like):
        """A second stream for chains of collectives on DIFFERENT communicators (each orders itself against the stream that is
        current when it is issued, not against the other communicator): GPUs only."""
        import torch
        if not like.is_cuda:
            return None
        if self._side is None:
            self._side = torch.cuda.Stream()
        return self._side


def distribute_grid_panel(grid, b_panel, b_src_p, mode="broadcast"):
    """Start moving sub-panel p of every column panel of B from rank 0 to the ranks that multiply with it; returns a handle with
    .wait().  b_panel: this rank's (l, w) buffer (on rank 0: b_src_p[0] itself); b_src_p: on rank 0 the Q source tensors
    [column panel j], None elsewhere.

    "broadcast": for every column panel j one broadcast from rank 0 inside {rank 0} + column group j.
    "allgather": ONE scatter from rank 0 over all ranks -- rank (i, j) receives rows [i l / R, (i + 1) l / R) of panel j: 1 / world of
    B leaves rank 0 towards every peer, all of its links in parallel -- then an all-gather inside every column group."""
    import torch
    import torch.distributed as dist
    R, Q, rank = grid.R, grid.Q, grid.rank
    if mode == "broadcast":
        works = []
        for j in range(Q):
            if rank == 0:
                works.append(dist.broadcast(b_src_p[j], src=0, group=grid.bgroups[j], async_op=True))
            elif j == grid.j:
                works.append(dist.broadcast(b_panel, src=0, group=grid.bgroups[j], async_op=True))
        return _Chain(works)
    assert b_panel.shape[0] % R == 0, "panel rows must divide by the number of row blocks"
    shards = list(b_panel.chunk(R, dim=0))
    mine = shards[grid.i]
    slist = None
    if rank == 0:
        slist = [list(b_src_p[grid.block_of(r)[1]].chunk(R, dim=0))[grid.block_of(r)[0]] for r in range(grid.world)]
    gloo = dist.get_backend() != "nccl"
    w1 = dist.scatter(mine, scatter_list=slist, src=0, async_op=True)
    if R == 1:
        return w1
    side = grid.side_stream(b_panel)
    if side is not None and Q > 1:
        # the all-gather runs on the column group's communicator: chained behind the scatter through a side stream, so that the
        # compute stream waits for neither until the consumer asks (handle.wait())
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            w1.wait()
            w2 = dist.all_gather(shards, mine, group=grid.col_groups[grid.j], async_op=True)
        return _Chain([w2])
    src = mine
    if gloo:  # gloo runs asynchronous operations in no particular order, and dislikes aliased buffers
        w1.wait()
        src = mine.clone()
    w2 = dist.all_gather(shards, src, group=grid.col_groups[grid.j], async_op=True)
    return _Chain([w1, w2])


def gather_grid_panel(grid, c_panel, c_full_p):
    """Start gathering sub-panel p of C on rank 0: c_panel = this rank's (rows / R, w) block; c_full_p = on rank 0 the Q tensors
    (rows, w) [column panel j] that receive it, None elsewhere.  One gather over all ranks."""
    import torch.distributed as dist
    glist = None
    if grid.rank == 0:
        glist = [list(c_full_p[grid.block_of(r)[1]].chunk(grid.R, dim=0))[grid.block_of(r)[0]] for r in range(grid.world)]
    return dist.gather(c_panel, gather_list=glist, dst=0, async_op=True)


def mul_grid_pipelined(grid, a_block, b_src, b_panels, c_panels, c_full, ncols_inner, ncols_sub, local_mul=None,
                       bcast="broadcast", **kw):
    """C = A * B on an R x Q grid of ranks, B in Q column panels of P sub-panels each.

    a_block        : (rows / R, ceil(l / 64)) int64, row block i of A
    b_src[j][p]    : rank 0 only -- (l, ceil(ncols_sub / 64)) int64, sub-panel p of column panel j of B; None elsewhere
    b_panels[p]    : this rank's sub-panels of ITS column panel j (rank 0: b_src[0][p])
    c_panels[p]    : (rows / R, ceil(ncols_sub / 64)), this rank's block of C for sub-panel p
    c_full[j][p]   : rank 0 only -- (rows, ceil(ncols_sub / 64)) receiving sub-panel p of column panel j of C; None elsewhere
    """
    local_mul = local_mul or _hip_local_mul
    P = len(b_panels)
    if grid.world == 1:
        for p in range(P):
            local_mul(a_block, b_panels[p], c_panels[p], ncols_inner, ncols_sub, **kw)
            if c_full is not None:
                c_full[0][p].copy_(c_panels[p])
        return
    bcasts = [distribute_grid_panel(grid, b_panels[p], [b_src[j][p] for j in range(grid.Q)] if grid.rank == 0 else None, bcast)
              for p in range(P)]
    gathers = []
    for p in range(P):
        bcasts[p].wait()
        local_mul(a_block, b_panels[p], c_panels[p], ncols_inner, ncols_sub, **kw)
        gathers.append(gather_grid_panel(grid, c_panels[p], [c_full[j][p] for j in range(grid.Q)] if grid.rank == 0 else None))
    for w in gathers:
        w.wait()


def step_grid(grid, A, b_src_t, Bp_t, Cp_t, Cfull_t, Bp, Cp, algo="auto", levels=0, stream=None, bcast="broadcast", events=None):
    """bench.py's timed step for N > 1 ranks on an R x Q grid (DMat wrappers pre-built); `events` as in step_pipelined."""
    import torch
    P = len(Bp_t)

    def mark():
        if events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            events.append(e)
    mark()
    bcasts = [distribute_grid_panel(grid, Bp_t[p], [b_src_t[j][p] for j in range(grid.Q)] if grid.rank == 0 else None, bcast)
              for p in range(P)]
    gathers = []
    for p in range(P):
        bcasts[p].wait()
        mark()
        device.mul(A, Bp[p], C=Cp[p], algo=algo, param=levels, stream=stream)
        mark()
        gathers.append(gather_grid_panel(grid, Cp_t[p], [Cfull_t[j][p] for j in range(grid.Q)] if grid.rank == 0 else None))
    for w in gathers:
        w.wait()
    mark()


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
