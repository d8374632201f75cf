"""Multi-GPU execution: branches and slices of the contraction tree sharded over ranks, one
collective at the join.

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


def contract_sliced(executor, arrays, strip_exponent=False, group=None, rank=None, world=None, slice_graph=None):
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
    out = executor(arrays, strip_exponent=strip_exponent, slices=mine, slice_graph=slice_graph)
    if strip_exponent:
        out, e = out
    if world == 1:
        return (out.to_numpy(), e) if strip_exponent else out.to_numpy()

    buf = out._buf
    is_dev_tensor = isinstance(buf, torch.Tensor)
    t = buf if is_dev_tensor else torch.from_numpy(np.ascontiguousarray(out.to_numpy()).reshape(-1).copy())
    n = max(out.size, 1)
    if strip_exponent and n <= (1 << 16):
        # ONE collective: every rank contributes (mantissa block, exponent) -- 8 (n + 1) bytes, 16 for an
        # amplitude -- and adds the gathered blocks up on the common (max) exponent itself
        # (complex outputs travel as interleaved (re, im) doubles: a cast to float64 would drop the imaginary part)
        cplx = t.is_complex()
        body = torch.view_as_real(t[:n].to(torch.complex128)).reshape(-1) if cplx else t[:n].to(torch.float64)
        nb = body.numel()
        mine = torch.empty(nb + 1, dtype=torch.float64, device=t.device)
        mine[:nb] = body
        mine[nb] = e if np.isfinite(e) else -1e300
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)
        g = torch.stack(gathered).cpu().numpy()       # world x (nb + 1) doubles: the result itself, read back once
        es = g[:, nb]
        e_max = float(es.max())
        blocks = g[:, :nb].reshape(world, n, 2) if cplx else g[:, :nb]
        blocks = blocks[..., 0] + 1j * blocks[..., 1] if cplx else blocks
        if e_max <= -1e299:
            return np.zeros(out.shape, dtype=blocks.dtype), 0.0
        res = (blocks * (10.0 ** (es - e_max))[:, None]).sum(0)
        return res.astype(out.to_numpy().dtype if hasattr(out, "to_numpy") else res.dtype).reshape(out.shape), e_max
    if strip_exponent:
        # large outputs: a MAX of the exponents, then a SUM of the rescaled mantissas (two collectives)
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


# ---------------------------------------------------------------------------
# branches x slices: the two half sweeps of a 2D network on two groups of ranks
# ---------------------------------------------------------------------------
def two_sided_layout(nslices, world):
    """Who evaluates what: ``[(branch, slice numbers)]`` per rank.  Ranks ``0 .. world//2 - 1`` sweep the top
    half, the others the bottom half; inside a group the slices of the cut row go out in CONTIGUOUS blocks
    (neighbouring slices share their prefix tensors, twosided.py).  One rank alone evaluates both branches."""
    if world == 1:
        return [("both", list(range(nslices)))]
    nt = world // 2
    nb = world - nt
    out = []
    for n, name in ((nt, "top"), (nb, "bottom")):
        for blk in np.array_split(np.arange(nslices), n):
            out.append((name, [int(s) for s in blk]))
    return out


def sliced_cols_for_world(sizes, world):
    """Fewest leading cut bonds whose slices give every rank of a branch group at least one."""
    need = max(world - world // 2, 1)
    k, n = 0, 1
    while n < need and k < len(sizes):
        n *= sizes[k]
        k += 1
    return k


def contract_two_sided(plan, arrays, strip_exponent=False, group=None, stats=None):
    """``plan``: a ``twosided.TwoSidedContraction``.  Every rank evaluates its (branch, block of slices): the rows
    of its half that no slice touches, then its slabs ``T[s]`` / ``B[s]`` of the cut row.  Join: the top rank
    owning slice s hands ``T[s]`` to the bottom rank owning it (point-to-point over xGMI, slab by slab, so the
    hand-off overlaps the remaining slabs), the bottom rank takes the dot products, and ONE collective -- an
    all-gather of every rank's (mantissa, exponent) pair, 16 bytes each -- ends the job; each rank adds the
    pairs up on a common exponent.  Returns the value (or ``(mantissa, exponent)``) on every rank."""
    import time

    import torch
    import torch.distributed as dist

    from .twosided import combine_pairs

    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    if world == 1:
        return plan(arrays, strip_exponent=strip_exponent)
    layout = two_sided_layout(plan.nslices, world)
    branch, mine = layout[rank]
    nt = world // 2
    owner_top = {s: r for r in range(nt) for s in layout[r][1]}
    owner_bot = {s: r for r in range(nt, world) for s in layout[r][1]}
    half = plan.top if branch == "top" else plan.bottom
    from .array import asarray

    arrays = [asarray(a) for a in arrays]
    dev = arrays[0]._dev
    t0 = time.perf_counter()
    U, eU = half.hoist(arrays)
    if stats is not None:
        dev.synchronize() if hasattr(dev, "synchronize") else None
        stats["hoist_s"] = time.perf_counter() - t0

    # gloo moves host memory only: with it (CPU tests; several ranks sharing one GPU as a debugging aid) messages
    # are staged through the host -- RCCL ("nccl") sends device buffers as they are
    host_staged = dist.get_backend(group) == "gloo"

    def as_tensor(x):
        buf = x._buf
        if isinstance(buf, torch.Tensor):
            t = buf[: max(x.size, 1)]
            return t.cpu() if host_staged else t
        return torch.from_numpy(np.ascontiguousarray(x.to_numpy()).reshape(-1).copy())

    pairs = []
    sent = []      # every slab handed over stays referenced until the job is over (the transport may still read it)
    if branch == "top":
        for s, T, eT in half.slabs(arrays, U, mine):
            dst = owner_bot[s]
            gdst = dst if group is None else dist.get_global_rank(group, dst)
            payload = as_tensor(T).contiguous()
            expo = torch.tensor([eT + eU], dtype=torch.float64, device=payload.device)
            # non-blocking: this rank goes on to its next slab while the transfer runs; everything handed over stays
            # referenced (``sent``) until the closing collective is behind it
            sent.append((T, payload, expo, dist.isend(payload, dst=gdst, group=group), dist.isend(expo, dst=gdst, group=group)))
        for item in sent:
            item[3].wait()
            item[4].wait()
    else:
        from .array import Array

        for s, B, eB in half.slabs(arrays, U, mine):
            src = owner_top[s]
            gsrc = src if group is None else dist.get_global_rank(group, src)
            tb = as_tensor(B)
            recv = torch.empty_like(tb)
            et = torch.zeros(1, dtype=torch.float64, device=tb.device)
            works = [dist.irecv(recv, src=gsrc, group=group), dist.irecv(et, src=gsrc, group=group)]
            for w_ in works:
                w_.wait()
            if isinstance(B._buf, torch.Tensor):
                T = Array(dev, recv.to(B._buf.device), B.shape, B.dtype)
            else:
                T = asarray(recv.numpy().reshape(B.shape))
            pairs.append(plan.join(T, float(et.cpu()[0]), B, eB + eU))
    m, e = combine_pairs(pairs, strip_exponent=True) if pairs else (0.0, float("-inf"))
    if stats is not None:
        dev.synchronize() if hasattr(dev, "synchronize") else None
        stats["compute_s"] = time.perf_counter() - t0
    # ---- the one collective of the job -------------------------------------------------------------------
    tdev = as_tensor(arrays[0]).device
    mine_t = torch.tensor([m, e if np.isfinite(e) else -1e300], dtype=torch.float64, device=tdev)
    gathered = [torch.empty_like(mine_t) for _ in range(world)]
    dist.all_gather(gathered, mine_t, group=group)
    allp = [(float(g[0]), float(g[1])) for g in (x.cpu() for x in gathered) if float(g[1]) > -1e299]
    del sent       # the all-gather's read-back above is behind every hand-off of this rank
    return combine_pairs(allp, strip_exponent=strip_exponent)
