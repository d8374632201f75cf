"""Multi-process tests of the multi-GPU path on CPU (gloo = the RCCL collectives' stand-in, device ops
interpreted by tests/emu_device.py): slices sharded round-robin with one collective at the join
(world 2, 4, 8, uneven slice counts, unsliced trees), and the branch decomposition -- the two half sweeps
of a 2D network on two groups of ranks, cut-row slices inside a group, point-to-point hand-off of the cut
boundary, one all-gather (world 2, 3, 4, 8)."""

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, strip, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import quimb_amd as qa
    import quimb_amd.device as qd
    from emu_device import EmuDevice
    from oracle import np_oracle as orc
    from quimb_amd.distributed import contract_sliced, rank_slices

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        qd.set_default_device(EmuDevice())
        arrays, inputs = orc.tn2d_rand(4, 4, 3, seed=21, dtype="float64")
        size = {ix: 3 for t in inputs for ix in t}
        tree = qa.ContractionTree(inputs, (), size, path=qa.sweep_path_2d(4, 4))
        st = qa.find_slices(tree, target_slices=9)
        mine = list(rank_slices(st.nslices, rank, world))
        assert mine and len(mine) < st.nslices
        ex = qa.TreeExecutor(st, "float64")
        out = contract_sliced(ex, arrays, strip_exponent=strip)
        val = out[0] * 10.0 ** out[1] if strip else out
        np.save(os.path.join(outdir, f"r{rank}.npy"), np.asarray(val))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("strip", [False, True])
def test_sliced_two_ranks_gloo(tmp_path, strip):
    import torch.multiprocessing as mp

    from oracle import np_oracle as orc

    port = _free_port()
    mp.spawn(_worker, args=(2, port, strip, str(tmp_path)), nprocs=2, join=True)
    arrays, inputs = orc.tn2d_rand(4, 4, 3, seed=21, dtype="float64")
    want = orc.oracle_array_contract(arrays, inputs, ())
    for r in range(2):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got.item() == pytest.approx(want.item(), rel=1e-10)


def _worker_sliced_uneven(rank, world, port, nslices_target, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import quimb_amd as qa
    import quimb_amd.device as qd
    from emu_device import EmuDevice
    from oracle import np_oracle as orc
    from quimb_amd.distributed import contract_sliced

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        qd.set_default_device(EmuDevice())
        arrays, inputs = orc.tn2d_rand(3, 4, 3, seed=5, dtype="float64")
        size = {ix: 3 for t in inputs for ix in t}
        tree = qa.ContractionTree(inputs, (), size, path=qa.sweep_path_2d(3, 4))
        if nslices_target > 1:
            tree = qa.find_slices(tree, target_slices=nslices_target)
        ex = qa.TreeExecutor(tree, "float64")
        m, e = contract_sliced(ex, arrays, strip_exponent=True)
        plain = contract_sliced(ex, arrays)
        np.save(os.path.join(outdir, f"r{rank}.npy"), np.asarray([np.asarray(m).item() * 10.0**e, np.asarray(plain).item()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nslices", [(4, 9), (8, 27), (8, 3), (2, 1), (4, 1)])
def test_sliced_uneven_and_unsliced(tmp_path, world, nslices):
    """Slice counts that do not divide the world size, more ranks than slices, and an UNSLICED tree (one slice:
    only rank 0 owns it -- every other rank must contribute zero, not the full value)."""
    import torch.multiprocessing as mp

    from oracle import np_oracle as orc

    mp.spawn(_worker_sliced_uneven, args=(world, _free_port(), nslices, str(tmp_path)), nprocs=world, join=True)
    arrays, inputs = orc.tn2d_rand(3, 4, 3, seed=5, dtype="float64")
    want = orc.oracle_array_contract(arrays, inputs, ()).item()
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got[0] == pytest.approx(want, rel=1e-10) and got[1] == pytest.approx(want, rel=1e-10)


def _worker_two_sided(rank, world, port, Lx, Ly, D, k, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import quimb_amd.device as qd
    from emu_device import EmuDevice
    from oracle import np_oracle as orc
    from quimb_amd.distributed import contract_two_sided, sliced_cols_for_world
    from quimb_amd.twosided import TwoSidedContraction

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        qd.set_default_device(EmuDevice())
        arrays, inputs = orc.tn2d_rand(Lx, Ly, D, seed=11, dtype="float64")
        size = {ix: D for t in inputs for ix in t}
        if k is None:
            k = sliced_cols_for_world([D] * Ly, world)
        plan = TwoSidedContraction(inputs, size, Lx, Ly, "float64", sliced_cols=k)
        stats = {}
        m, e = contract_two_sided(plan, arrays, strip_exponent=True, stats=stats)
        plain = contract_two_sided(plan, arrays)
        assert "compute_s" in stats
        np.save(os.path.join(outdir, f"r{rank}.npy"), np.asarray([m * 10.0**e, plain]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,Lx,Ly,D,k", [(2, 4, 4, 3, None), (2, 4, 4, 3, 1), (3, 5, 3, 2, None), (4, 4, 4, 3, None),
                                              (4, 4, 3, 2, 2), (8, 4, 4, 2, None), (8, 3, 4, 3, 1)])
def test_two_sided_branches(tmp_path, world, Lx, Ly, D, k):
    """Top / bottom half sweeps on two groups of ranks, the cut row's slices in contiguous blocks inside a
    group (uneven blocks, ranks without a slice, odd world sizes), slabs handed over point to point, one
    all-gather: every rank ends with the oracle's value."""
    import torch.multiprocessing as mp

    from oracle import np_oracle as orc

    mp.spawn(_worker_two_sided, args=(world, _free_port(), Lx, Ly, D, k, str(tmp_path)), nprocs=world, join=True)
    arrays, inputs = orc.tn2d_rand(Lx, Ly, D, seed=11, dtype="float64")
    want = orc.oracle_array_contract(arrays, inputs, ()).item()
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got[0] == pytest.approx(want, rel=1e-10) and got[1] == pytest.approx(want, rel=1e-10)


# ---------------------------------------------------------------------------------------------------------
# quadrant tree sharded over ranks: blocks of the two joins (range-sliced cut bonds), one all-gather
# ---------------------------------------------------------------------------------------------------------
def _worker_quadrants(rank, world, port, Lx, Ly, D, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import quimb_amd.device as qd
    from emu_device import EmuDevice
    from oracle import np_oracle as orc
    from quimb_amd.quadrants import QuadrantRank, QuadrantSharding, contract_quadrants

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        qd.set_default_device(EmuDevice())
        arrays, inputs = orc.tn2d_rand(Lx, Ly, D, seed=33, dtype="float64")
        size = {ix: D for t in inputs for ix in t}
        sh = QuadrantSharding(inputs, size, Lx, Ly, world)
        assert sh.P * sh.Q == world
        plan = QuadrantRank(sh, rank, "float64")
        local = sh.shard(arrays, rank)
        # the rank's network really is smaller: its sliced bonds keep 1 / parts of their values
        assert sum(a.size for a in local) < sum(a.size for a in arrays)
        val = contract_quadrants(plan, local)
        m, e = contract_quadrants(plan, local, strip_exponent=True)
        np.save(os.path.join(outdir, f"r{rank}.npy"), np.asarray([val, m * 10.0**e]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,Lx,Ly,D", [(2, 4, 4, 4), (4, 4, 4, 4), (8, 4, 6, 2), (3, 4, 4, 3), (6, 4, 4, 6)])
def test_quadrants_sharded_gloo(tmp_path, world, Lx, Ly, D):
    """Every rank contracts its block of T and B (the same tree on a range-sliced network) and ONE all-gather of
    (mantissa, exponent) pairs ends the job: the oracle's value of the whole network on every rank, at world
    2 / 4 / 8 (halves of one / two / three cut bonds) and for rank counts with a factor 3 (thirds of a bond)."""
    import torch.multiprocessing as mp

    from oracle import np_oracle as orc

    port = _free_port()
    mp.spawn(_worker_quadrants, args=(world, port, Lx, Ly, D, str(tmp_path)), nprocs=world, join=True)
    arrays, inputs = orc.tn2d_rand(Lx, Ly, D, seed=33, dtype="float64")
    want = np.asarray(orc.oracle_array_contract(arrays, inputs, ())).item()
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got[0] == pytest.approx(want, rel=1e-10) and got[1] == pytest.approx(want, rel=1e-10)


def test_quadrant_sharding_costs():
    """The headline instance (10 x 10, D = 6): grids 2x1 / 2x2 / 4x2 on halves of 1 / 2 / 3 cut bonds, the busiest rank
    at no more than 1/5 of the one-rank multiplications on 8 ranks, every rank's joins (7776 / P) x (7776 / Q) x 7776."""
    from oracle import np_oracle as orc
    import quimb_amd as qa
    from quimb_amd.quadrants import QuadrantSharding

    _, inputs = orc.tn2d_rand(10, 10, 2, seed=0)
    size = {ix: 6 for t in inputs for ix in t}
    for world, grid in ((1, [1, 1]), (2, [2, 1]), (4, [2, 2]), (8, [4, 2])):
        sh = QuadrantSharding(inputs, size, 10, 10, world)
        rep = sh.cost_report()
        assert rep["grid"] == grid and all(p == 2 for p in rep["parts_per_bond"])
        assert rep["ideal_speedup_vs_one_rank"] > 0.88 * world
        ex = qa.TreeExecutor(sh.tree(world - 1), "float32")
        joins = sorted((i.M * i.N * i.K for i in ex.info), reverse=True)[:2]
        assert joins == [7776**3 // world] * 2
    assert QuadrantSharding(inputs, size, 10, 10, 8).cost_report()["busiest_rank_fraction"] <= 0.2
    # every rank keeps a different block and together they cover the cut bonds exactly once
    sh = QuadrantSharding(inputs, size, 10, 10, 8)
    seen = set()
    for r in range(8):
        seen.add(tuple(sorted((repr(b), rng) for b, rng in sh.rank_ranges(r).items())))
    assert len(seen) == 8


# ---------------------------------------------------------------------------------------------------------
# RCCL itself.  Two ranks: runs by default wherever >= 2 GPUs are visible (RCCL refuses two ranks on one device, so the
# one-GPU boxes of the test tier skip it).  ONE rank: runs on every GPU box -- RCCL initialises, and the job's closing
# all-gather (quadrants.contract_quadrants) plus an all-reduce execute on device buffers in a group of one, so the first
# multi-GPU bench run is not the first time this process environment (HSA_ENABLE_IPC_MODE_LEGACY=0) meets the library.
# ---------------------------------------------------------------------------------------------------------
def _worker_nccl(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        import quimb_amd as qa
        from oracle import np_oracle as orc
        from quimb_amd.distributed import contract_sliced
        from quimb_amd.quadrants import QuadrantRank, QuadrantSharding, contract_quadrants

        arrays, inputs = orc.tn2d_rand(6, 6, 6, seed=4, dtype="float32")
        size = {ix: 6 for t in inputs for ix in t}
        sh = QuadrantSharding(inputs, size, 6, 6, world)
        plan = QuadrantRank(sh, rank, "float32")
        local = sh.shard([qa.asarray(a) for a in arrays], rank)
        m, e = contract_quadrants(plan, local, strip_exponent=True)
        tree = qa.find_slices(qa.ContractionTree(inputs, (), size, path=qa.sweep_path_2d(6, 6)), target_slices=6)
        ms, es = contract_sliced(qa.TreeExecutor(tree, "float32"), arrays, strip_exponent=True)
        np.save(os.path.join(outdir, f"r{rank}.npy"), np.asarray([m * 10.0**e, float(np.asarray(ms).reshape(-1)[0]) * 10.0**es]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_nccl_two_ranks_smoke(tmp_path):
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL does not share a device between ranks); the one-rank RCCL test covers this box")
    import torch.multiprocessing as mp

    from oracle import np_oracle as orc

    port = _free_port()
    mp.spawn(_worker_nccl, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    arrays, inputs = orc.tn2d_rand(6, 6, 6, seed=4, dtype="float64")
    want = np.asarray(orc.oracle_array_contract(arrays, inputs, ())).item()
    for r in range(2):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got[0] == pytest.approx(want, rel=2e-6) and got[1] == pytest.approx(want, rel=2e-6)


def _worker_nccl_one(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import quimb_amd as qa
        from oracle import np_oracle as orc
        from quimb_amd.quadrants import QuadrantRank, QuadrantSharding, contract_quadrants

        assert dist.get_backend() == "nccl"
        arrays, inputs = orc.tn2d_rand(6, 6, 6, seed=4, dtype="float32")
        size = {ix: 6 for t in inputs for ix in t}
        sh = QuadrantSharding(inputs, size, 6, 6, 1)
        plan = QuadrantRank(sh, 0, "float32")
        local = sh.shard([qa.asarray(a) for a in arrays], 0)
        m, e = contract_quadrants(plan, local, strip_exponent=True)      # ends in dist.all_gather on device doubles
        t = torch.arange(4, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        mx = torch.tensor([3.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.barrier()
        np.save(os.path.join(outdir, "r0.npy"), np.asarray([m * 10.0**e, float(t.sum().cpu()), float(mx.cpu()[0])]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_nccl_one_rank_collectives(tmp_path):
    """RCCL on the one-GPU box: process group of one rank, the job's closing all-gather on device buffers."""
    import torch.multiprocessing as mp

    from oracle import np_oracle as orc

    mp.spawn(_worker_nccl_one, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    arrays, inputs = orc.tn2d_rand(6, 6, 6, seed=4, dtype="float64")
    want = np.asarray(orc.oracle_array_contract(arrays, inputs, (), path=None)).item()
    got = np.load(tmp_path / "r0.npy")
    assert got[0] == pytest.approx(want, rel=2e-6) and got[1] == 6.0 and got[2] == 3.5


def _worker_sliced_complex(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import quimb_amd as qa
    import quimb_amd.device as qd
    from emu_device import EmuDevice
    from oracle import np_oracle as orc
    from quimb_amd.distributed import contract_sliced

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        qd.set_default_device(EmuDevice())
        arrays, inputs = orc.tn2d_rand(3, 3, 3, seed=9, dtype="float64")
        rng = np.random.default_rng(9)
        arrays = [(a + 1j * rng.uniform(-0.5, 0.5, size=a.shape)).astype("complex128") for a in arrays]
        size = {ix: 3 for t in inputs for ix in t}
        st = qa.find_slices(qa.ContractionTree(inputs, (), size, path=qa.sweep_path_2d(3, 3)), target_slices=3)
        m, e = contract_sliced(qa.TreeExecutor(st, "complex128"), arrays, strip_exponent=True)
        np.save(os.path.join(outdir, f"r{rank}.npy"), np.asarray(m).reshape(-1)[:1] * 10.0**e)
    finally:
        dist.destroy_process_group()


def test_sliced_complex_strip_exponent_gloo(tmp_path):
    """A COMPLEX network through the one-collective join of ``contract_sliced`` (round 2 cast the gathered mantissa to
    float64 and lost the imaginary part): real and imaginary part of the oracle's value on both ranks."""
    import torch.multiprocessing as mp

    from oracle import np_oracle as orc

    port = _free_port()
    mp.spawn(_worker_sliced_complex, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    arrays, inputs = orc.tn2d_rand(3, 3, 3, seed=9, dtype="float64")
    rng = np.random.default_rng(9)
    arrays = [(a + 1j * rng.uniform(-0.5, 0.5, size=a.shape)).astype("complex128") for a in arrays]
    want = complex(np.asarray(orc.oracle_array_contract(arrays, inputs, ())).item())
    assert abs(want.imag) > 1e-6 * abs(want)
    for r in range(2):
        got = complex(np.load(tmp_path / f"r{r}.npy")[0])
        assert abs(got - want) <= 1e-10 * abs(want), (got, want)


def _worker_blocks_complex(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import quimb_amd as qa
    import quimb_amd.device as qd
    from emu_device import EmuDevice
    from quimb_amd.quadrants import QuadrantRank, QuadrantSharding, contract_quadrants
    from quimb_amd.rangeslice import RangeSliced, RangeSlicedExecutor, contract_range_sliced, find_range_slices

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        qd.set_default_device(EmuDevice())
        arrays, inputs, size = _complex_lattice()
        quad = qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(4, 4))
        rse = RangeSlicedExecutor(RangeSliced(quad, find_range_slices(quad, 4)), "complex128")
        m1, e1 = contract_range_sliced(rse, arrays, strip_exponent=True)
        v1 = contract_range_sliced(rse, arrays)
        sh = QuadrantSharding(inputs, size, 4, 4, world)
        plan = QuadrantRank(sh, rank, "complex128")
        m2, e2 = contract_quadrants(plan, sh.shard(arrays, rank), strip_exponent=True)
        v2 = contract_quadrants(plan, sh.shard(arrays, rank))
        np.save(os.path.join(outdir, f"r{rank}.npy"),
                np.asarray([m1 * 10.0**e1, np.asarray(v1).item(), m2 * 10.0**e2, v2], dtype=np.complex128))
    finally:
        dist.destroy_process_group()


def _complex_lattice():
    from oracle import np_oracle as orc

    arrays, inputs = orc.tn2d_rand(4, 4, 4, seed=21, dtype="float64")
    rng = np.random.default_rng(21)
    arrays = [(a + 1j * rng.uniform(-0.5, 0.5, size=a.shape)).astype("complex128") for a in arrays]
    inputs = [tuple(t) for t in inputs]
    return arrays, inputs, {ix: 4 for t in inputs for ix in t}


def test_range_sliced_and_quadrants_complex_gloo(tmp_path):
    """the joins of ``contract_range_sliced`` / ``contract_quadrants`` carried the mantissa as
    ONE float64 -- a complex network raised TypeError in the first and silently lost its imaginary part in the second.
    Both now gather (re, im, exponent); real and imaginary part of the oracle's value on both ranks."""
    import torch.multiprocessing as mp

    from oracle import np_oracle as orc

    port = _free_port()
    mp.spawn(_worker_blocks_complex, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    arrays, inputs, _ = _complex_lattice()
    want = complex(np.asarray(orc.oracle_array_contract(arrays, inputs, ())).item())
    assert abs(want.imag) > 1e-6 * abs(want)
    for r in range(2):
        for got in np.load(tmp_path / f"r{r}.npy"):
            assert abs(complex(got) - want) <= 1e-10 * abs(want), (got, want)
