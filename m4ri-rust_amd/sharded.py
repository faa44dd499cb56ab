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
    _lib.check(_lib.lib().gf2_dmat_fill_random_rows(ctypes.byref(A.s), seed, row0, stream), "gf2_dmat_fill_random_rows")


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
