"""world_size-2 test of the multi-GPU path on CPU: slices sharded round-robin
over two gloo ranks, one all-reduce at the join (the RCCL collective's stand-in),
device ops interpreted by tests/emu_device.py."""

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
