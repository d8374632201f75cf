"""``quimb_amd.Options``: one explicit, immutable object instead of environment switches (quimb_amd/options.py) -- the
per-thread default stack (as quimb keeps its contraction defaults, quimb/tensor/contraction.py:23-35), what captures it and
when, and that nothing re-reads the environment after import."""
import os
import threading

import numpy as np
import pytest

import quimb_amd as qa
from quimb_amd.options import Options


def _chain(n=5, d=8):
    inputs = [(f"i{k}", f"i{k + 1}") for k in range(n)]
    size = {ix: d for t in inputs for ix in t}
    return inputs, size


def test_defaults_stack_and_scope():
    base = qa.get_options()
    assert isinstance(base, Options) and base.lanes and base.fuse_pairs and base.chain2_kernel == "auto"
    with qa.exec_options(lanes=False, join_dot=False) as o:
        assert qa.get_options() is o and not o.lanes and not o.join_dot and o.fuse_pairs
        with qa.exec_options(fuse_pairs=False):
            assert not qa.get_options().fuse_pairs and not qa.get_options().lanes
        assert qa.get_options() is o
    assert qa.get_options() is base
    old = qa.set_options(debug=True)
    try:
        assert old is base and qa.get_options().debug
    finally:
        qa.set_options(debug=False)
    with pytest.raises(Exception):
        base.lanes = False                      # frozen
    with pytest.raises(TypeError):
        base.replace(no_such_option=1)


def test_default_is_per_thread():
    seen = {}

    def worker():
        seen["inside"] = qa.get_options().lanes
        with qa.exec_options(lanes=False):
            seen["scoped"] = qa.get_options().lanes

    with qa.exec_options(lanes=False):
        th = threading.Thread(target=worker)
        th.start()
        th.join()
        assert not qa.get_options().lanes
    assert seen == {"inside": True, "scoped": False}      # the other thread saw the process default, not this scope


def test_executors_and_expressions_capture_options_when_built(emu):
    inputs, size = _chain()
    tree = qa.find_path(inputs, ("i0", "i5"), size, "greedy")
    with qa.exec_options(regroup=False, fuse_pairs=False):
        ex = qa.TreeExecutor(tree, "float64")
    assert not ex.options.fuse_pairs and not ex.options.regroup          # kept after the scope closed
    assert qa.TreeExecutor(tree, "float64").options.fuse_pairs
    ex2 = qa.TreeExecutor(tree, "float64", options=qa.get_options().replace(lanes=False))
    assert not ex2.options.lanes
    # an expression is cached per options: two scopes, two expressions; the same scope, the same object
    shapes = [(8, 8)] * 5
    e1 = qa.array_contract_expression(inputs, ("i0", "i5"), shapes=shapes, optimize="greedy", dtype="float64")
    with qa.exec_options(auto_program=False):
        e2 = qa.array_contract_expression(inputs, ("i0", "i5"), shapes=shapes, optimize="greedy", dtype="float64")
        assert qa.array_contract_expression(inputs, ("i0", "i5"), shapes=shapes, optimize="greedy", dtype="float64") is e2
    assert e1 is not e2 and e1.options.auto_program and not e2.options.auto_program
    assert qa.array_contract_expression(inputs, ("i0", "i5"), shapes=shapes, optimize="greedy", dtype="float64") is e1
    rng = np.random.default_rng(0)
    mats = [rng.normal(size=(8, 8)) for _ in range(5)]
    want = np.linalg.multi_dot(mats)
    for e in (e1, e2):
        np.testing.assert_allclose(np.asarray(e(*mats)), want, rtol=1e-12)
    # per-call behaviour is a call argument
    np.testing.assert_allclose(ex(mats, lanes=False).to_numpy(), want, rtol=1e-12)


def test_environment_is_read_once_at_import(monkeypatch):
    """``Options.from_env`` maps the old switch names; flipping the environment later changes nothing."""
    o = Options.from_env({"QAMD_LANES": "0", "QAMD_CHAIN2": "0", "QAMD_CHAIN2Q": "2", "QAMD_HOLD_LATE": "1",
                          "QAMD_AUTO_PROGRAM_MAX_BYTES": "123", "QAMD_TILE_CFG": "6"})
    assert (o.lanes, o.fuse_pairs, o.chain2_kernel, o.hold_late, o.auto_program_max_bytes, o.tile_cfg) == \
        (False, False, "quad", "1", 123, 6)
    assert Options.from_env({"QAMD_CHAIN2R": "0"}).chain2_kernel == "lds" and Options.from_env({}).__eq__(Options())
    before = qa.get_options()
    monkeypatch.setenv("QAMD_LANES", "0")
    monkeypatch.setenv("QAMD_CHAIN2", "0")
    assert qa.get_options() is before and before.lanes
    inputs, size = _chain()
    assert qa.TreeExecutor(qa.find_path(inputs, ("i0", "i5"), size, "greedy"), "float64").options.lanes
