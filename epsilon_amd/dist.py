"""Multi-GPU plumbing: one process per GPU (torchrun), RCCL over xGMI for the data-path
collective, torch.distributed only for rendezvous.

Sharding model (include/epsilon_hip.h, "sharded solves"): every rank builds its LOCAL problem -
sharded variables and the elementwise constraint rows tied to them hold this rank's slice, the
data matrix holds this rank's column slab - and declares the sharded keys.  For the compiled
lasso the variables x, y and the row constraint:0 are sharded by columns of A;
the prox argument row arg:0 = A x - b (m entries) is replicated and is the one all-reduce per
sweep.  The iterates are those of the single-GPU solve up to the summation order of that
reduction (SURVEY.md 8(e) mode E1).
"""

import numpy as np

from . import _solve


def column_range(n, rank, world, align=64):
    """Contiguous, `align`-aligned split of n columns over `world` ranks."""
    per = -(-n // world)
    per = -(-per // align) * align
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


def init_comm(rank, world, backend="rccl", peer=False, slot_floats=0):
    """Collective: set up the solver library's communicator.  Needs torch.distributed to be
    initialised (any backend) for the rendezvous.  peer=True also tries to set up the one-shot
    peer-write window for the per-sweep messages; returns (enabled, reason)."""
    import torch.distributed as dist
    if backend == "rccl":
        obj = [_solve.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(obj, src=0)
        _solve.comm_init_rccl(rank, world, obj[0])
    elif backend == "host":
        import torch

        def allreduce(arr):
            t = torch.from_numpy(arr)
            if dist.get_backend() == "nccl":
                g = t.cuda()
                dist.all_reduce(g)
                t.copy_(g.cpu())
            else:
                dist.all_reduce(t)
        _solve.comm_init_callback(rank, world, allreduce)
    else:
        raise ValueError(backend)
    if peer:
        return _solve.comm_enable_peer(slot_floats)
    return False, "not requested"


def lasso_sharded_keys(prob):
    """Keys of the compiled lasso that are sharded by columns: both variables and the
    elementwise consensus row."""
    from . import ir
    keys = list(ir.get_variables(prob.proto()))
    keys += ["constraint:%d" % i for i in range(len(prob.constraints))]
    return keys


def mark_sharded(solver, prob):
    _solve.shard_keys(lasso_sharded_keys(prob))


def gather_variable(local_bytes, world):
    """All-gather the per-rank slices of a sharded variable (host side, for result checks)."""
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, local_bytes)
    return b"".join(parts)
