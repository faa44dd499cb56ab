#!/usr/bin/env python3
"""Every torch.distributed call of the multi-GPU step through RCCL itself, with the one rank a one-GPU box can hold (development tool;
no run on two devices has been possible for this repository -- this at least puts the collectives' RCCL code paths, their dtype, their
stream ordering against the product's stream and their aliasing of source and destination buffers under a real run):
    python3 tools/rccl_one_rank.py
world_size 1, backend nccl (= RCCL): broadcast, scatter + all-gather, gather as asynchronous operations on the compute stream, the
products between them by the library on the same stream, C compared with the product of the whole operands."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import gf2util as g
import m4ri_rust_amd  # noqa: F401
from m4ri_rust_amd import device, sharded

m = l = n = 4096
a, b = g.random_words(m, l, 1), g.random_words(l, n, 2)
comp = torch.cuda.Stream()
torch.cuda.set_stream(comp)
stream = comp.cuda_stream
A_t = torch.from_numpy(a.view(np.int64)).cuda()
A = device.DMat.from_torch(A_t, l)
ref = device.mul(A, device.DMat.from_torch(torch.from_numpy(b.view(np.int64)).cuda(), n), stream=stream).to_words()
grid = sharded.Grid(1, 1)
ok = True
for P in (1, 2, 4):
    ws, ncs = (n // 64) // P, n // P
    for bcast in ("broadcast", "allgather"):
        b_src_t = [[torch.from_numpy(b[:, p * ws:(p + 1) * ws].copy().view(np.int64)).cuda() for p in range(P)]]
        Bp_t = b_src_t[0]  # rank 0 multiplies out of its source panels
        Cp_t = [torch.zeros((m, ws), dtype=torch.int64, device="cuda") for _ in range(P)]
        Cfull_t = [[torch.zeros((m, ws), dtype=torch.int64, device="cuda") for _ in range(P)]]
        Bp = [device.DMat.from_torch(t, ncs) for t in Bp_t]
        Cp = [device.DMat.from_torch(t, ncs) for t in Cp_t]
        for _ in range(3):
            ev = []
            sharded.step_grid(grid, A, b_src_t, Bp_t, Cp_t, Cfull_t, Bp, Cp, stream=stream, bcast=bcast, events=ev)
        torch.cuda.synchronize()
        got = np.concatenate([Cfull_t[0][p].cpu().numpy().view(np.uint64) for p in range(P)], axis=1)
        same = np.array_equal(got, ref)
        ok = ok and same
        print("step_grid 1x1 P=%d %-9s via %s: %s (breakdown: %s)" % (P, bcast, dist.get_backend(), "C matches" if same else "MISMATCH",
                                                                   sharded.breakdown([ev])), flush=True)
        # the row-block step of earlier rounds (step_pipelined) through the same transports
        for t in Cp_t + Cfull_t[0]:
            t.zero_()
        sharded.step_pipelined(A, Bp_t, Cp_t, Cfull_t[0], Bp, Cp, stream=stream, bcast=bcast)
        torch.cuda.synchronize()
        got = np.concatenate([Cfull_t[0][p].cpu().numpy().view(np.uint64) for p in range(P)], axis=1)
        same = np.array_equal(got, ref)
        ok = ok and same
        print("step_pipelined  P=%d %-9s: %s" % (P, bcast, "C matches" if same else "MISMATCH"), flush=True)
# the timing reduction and the barrier of bench.py
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
info = [None]
dist.all_gather_object(info, {"rank": 0, "device": torch.cuda.current_device()})  # (the per-rank records of the N > 1 bench line)
print("all_reduce(MAX) / barrier / all_gather_object:", float(t[0]), info)
dist.destroy_process_group()
print("rccl_one_rank:", "ok" if ok else "FAILED")
sys.exit(0 if ok else 1)
