"""Multi-GPU execution: slices of the contraction tree sharded over ranks, one
reduce at the join.

The reference has no multi-device contraction: cotengra
slices are evaluated serially and summed (``Circuit.xeb_ex`` chunk map-reduce,
quimb/circuit/exact.py:1999-2018; ``cut_iter`` quimb/tensor/tensor_core.py:9291-9328).
Here rank ``r`` of ``W`` evaluates slices ``{s : s % W == r}`` -- independent units,
no data-path communication -- and the partial outputs are summed with a single
``all_reduce`` (RCCL over xGMI when the backend is ``nccl``; ``gloo`` in the CPU
tests).  The payload is the contraction *output* (a scalar for an amplitude), so
the collective is latency-bound and needs no bucketing.
"""

import numpy as np


def rank_slices(nslices, rank, world):
    """Round-robin slice ownership."""
    return range(rank, nslices, world)


def contract_sliced(executor, arrays, strip_exponent=False, group=None, rank=None, world=None):
    """Evaluate this rank's slices and all-reduce the result.

    Returns a numpy array on every rank (``(mantissa, exponent)`` if
    ``strip_exponent``).  Works without an initialised process group (world 1)."""
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        rank = dist.get_rank(group) if rank is None else rank
        world = dist.get_world_size(group) if world is None else world
    else:
        rank, world = 0, 1
    nsl = executor.tree.nslices
    mine = list(rank_slices(nsl, rank, world))
    out = executor(arrays, strip_exponent=strip_exponent, slices=mine)
    if strip_exponent:
        out, e = out
    if world == 1:
        return (out.to_numpy(), e) if strip_exponent else out.to_numpy()

    buf = out._buf
    is_dev_tensor = isinstance(buf, torch.Tensor)
    t = buf if is_dev_tensor else torch.from_numpy(np.ascontiguousarray(out.to_numpy()).reshape(-1).copy())
    n = max(out.size, 1)
    if strip_exponent:
        # bring every rank's mantissa to the common (max) exponent, then sum
        et = torch.tensor([e if np.isfinite(e) else -1e300], dtype=torch.float64, device=t.device)
        dist.all_reduce(et, op=dist.ReduceOp.MAX, group=group)
        e_max = float(et.cpu()[0])
        scale = 0.0 if not np.isfinite(e) else 10.0 ** (e - e_max)
        t = t[:n] * scale
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        res = t.cpu().numpy().reshape(out.shape)
        return res, e_max
    t = t[:n].clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy().reshape(out.shape)
