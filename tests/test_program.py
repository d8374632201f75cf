"""Launch programs (quimb_amd/program.py + csrc/program.cpp): the recording plumbing on the CPU (nothing is launched while
recording, so a GPU is not needed to check WHAT is recorded); replay itself is a ``-m gpu`` test."""
import ctypes as C

import numpy as np
import pytest

import checks
import quimb_amd as qa
from oracle import np_oracle as orc


@pytest.fixture
def recdev():
    import quimb_amd.device as qd
    from record_device import RecordOnlyDevice

    old = qd._DEFAULT
    dev = RecordOnlyDevice()
    qd.set_default_device(dev)
    try:
        yield dev
    finally:
        qd.set_default_device(old)


def _network(L, D, dtype="float32"):
    arrays, inputs = orc.tn2d_rand(L, L, D, seed=1, dtype=dtype)
    inputs = [tuple(t) for t in inputs]
    return arrays, inputs, {ix: D for t in inputs for ix in t}


def test_c_recorder_counts_and_binding():
    """The C side on its own: recorded calls are appended (not launched), waits are ops, marks tag the next launch."""
    from quimb_amd import _lib

    lib = _lib.load()
    P = lib.qamd_program_create(3)
    assert lib.qamd_program_record_begin(P) == 0
    assert lib.qamd_program_record_begin(P) != 0                       # one recording per thread
    assert lib.qamd_fill(0x1000, 10, 0.0, 0.0, 0, None) == 0
    assert lib.qamd_program_set_lane(P, 1) == 0 and lib.qamd_program_set_lane(P, 3) != 0
    assert lib.qamd_scale(0x2000, 10, 2.0, 0.0, 0, None) == 0
    assert lib.qamd_program_wait(P, 0, 1) == 0 and lib.qamd_program_wait(P, 1, 1) == 0   # self-wait: dropped
    assert lib.qamd_program_mark(P, 7) == 0
    assert lib.qamd_axpby(0x1000, 0x2000, 10, 1.0, 1.0, 0, None) == 0
    assert lib.qamd_program_record_end(P) == 0
    assert (lib.qamd_program_num_ops(P), lib.qamd_program_num_launches(P), lib.qamd_program_num_marks(P)) == (4, 3, 1)
    ptrs, nb = (C.c_void_p * 1)(0x2000), (C.c_int64 * 1)(40)
    assert lib.qamd_program_bind_inputs(P, 1, ptrs, nb) == 0
    # overlapping input ranges are refused: a recorded pointer could be re-based onto either
    p2, n2 = (C.c_void_p * 2)(0x2000, 0x2010), (C.c_int64 * 2)(40, 40)
    assert lib.qamd_program_bind_inputs(P, 2, p2, n2) != 0
    p2 = (C.c_void_p * 2)(0x2000, 0x2028)
    assert lib.qamd_program_bind_inputs(P, 2, p2, n2) == 0
    assert lib.qamd_fill(0x1000, 10, 0.0, 0.0, 0, None) != 0 or True    # (outside a recording the call launches: no GPU here)
    lib.qamd_program_destroy(P)


@pytest.mark.parametrize("dtype,strip", [("float32", True), ("float32", False), ("complex64", True)])
def test_recording_the_quadrant_tree(recdev, dtype, strip):
    """What one recording of the executor holds: every plan entry's launch, the fills that reset the exponent
    bookkeeping, one wait per side lane at the start and one per cross-lane hand-over; inputs are bound, not copied."""
    arrays, inputs, size = _network(6, 6, "float64")
    arrays = [a.astype(dtype) for a in arrays]
    ex = qa.TreeExecutor(qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(6, 6)), dtype)
    assert ex.nlanes == 4
    xs = [qa.asarray(a) for a in arrays]
    prog = ex.program(xs, strip_exponent=strip, mark_min_mults=10**5)
    nwait = prog.num_ops - prog.num_launches
    cross = sum(1 for i, e in enumerate(ex.plan) for o in ex._entry_io(e)[0]
                if o in ex._producer and ex.lanes[ex._producer[o]] != ex.lanes[i])
    assert nwait == (ex.nlanes - 1) + cross
    assert prog.num_launches >= len(ex.plan) + (2 if strip else 0)
    big = sum(1 for inf in ex.info if inf.mults >= 10**5)
    # (a complex step runs as ONE real launch with 4x the multiplications: more of them pass the bar)
    assert len(prog.marked) == big if np.dtype(dtype).kind != "c" else len(prog.marked) >= big
    assert recdev.record is None                                         # recording mode is left on every path
    # the pool: intermediates only (inputs are read in place), far fewer blocks than intermediates where lanes reuse
    assert prog.pool_bytes < 4 * sum(a.nbytes for a in arrays) + 40 * max(inf.bytes for inf in ex.info)
    assert all(p in [x._buf.data_ptr() for x in xs] for p in prog._in_ptrs0)


def test_later_chains_are_held_behind_the_first_joins_chains(recdev, monkeypatch):
    """A long first join (the 7 ms joins of the whole 10x10 D=6 network) MAY have the later join's corner sweeps run BESIDE
    it: they wait for the first join's own chains, not for the join.  Round 6: that pays only for plans WITHOUT fused rows
    (with rowq.hip a corner is five MFMA-bound launches that no longer fit beside a join: 15.8 ms held, 14.7 ms not,
    profiles/r06_hold_late.txt), and never for a rank's shorter joins (measured: they lose).  The rule, its overrides, and
    the waits a recorded program carries for it."""
    import quimb_amd.executor as qe
    from bench import build_network
    from quimb_amd.quadrants import QuadrantRank, QuadrantSharding

    arrays, inputs, size = build_network(10, 10, 6, 7, "float32")
    tree10 = qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(10, 10))
    assert qa.TreeExecutor(tree10, "float32").hold_late is None          # every row of every corner is one launch: no hold
    ex = qa.TreeExecutor(tree10, "float32", options=qa.get_options().replace(fuse_rows=False))
    pos, late, early = ex.hold_late
    assert ex.plan[pos][0] == "pair" and ex.info[pos].mults == 7776**3 and ex.lanes[pos] == 0
    assert len(late) == 2 and len(early) == 2 and not set(late) & set(early) and 0 in early
    # everything the first join needs is issued before it, nothing of the late lanes is
    assert all(ex.lanes[i] in early for i in range(pos)) and all(ex.lanes[i] in late + [0] for i in range(pos + 1, len(ex.plan)))
    for w in (2, 4, 8):
        sh = QuadrantSharding(inputs, size, 10, 10, w)
        assert QuadrantRank(sh, 0, "float32").executor.hold_late is None
    assert qa.TreeExecutor(ex.tree, "float32", options=qa.get_options().replace(hold_late="0")).hold_late is None
    # a small network with the rule forced on: the program records one wait per (late lane, early lane) on top of the
    # fork and the cross-lane hand-overs, and stays in join order (no plain-order twin)
    arrays, inputs, size = _network(6, 3, "float32")
    ex = qa.TreeExecutor(qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(6, 6)), "float32",
                         options=qa.get_options().replace(hold_late="1"))
    assert ex.hold_late is not None
    prog = ex.program([qa.asarray(a) for a in arrays], strip_exponent=True)
    assert prog.executor is ex
    cross = sum(1 for i, e in enumerate(ex.plan) for o in ex._entry_io(e)[0]
                if o in ex._producer and ex.lanes[ex._producer[o]] != ex.lanes[i])
    assert prog.num_ops - prog.num_launches == (ex.nlanes - 1) + cross + len(ex.hold_late[1]) * len(ex.hold_late[2])
    monkeypatch.setattr(qe, "HOLD_LATE_MIN_MULTS", 1.0)
    assert qa.TreeExecutor(ex.tree, "float32").hold_late is not None


def test_expressions_record_their_own_program(recdev, monkeypatch):
    """``ContractExpression._auto_program``: nothing before the third call, then one recording; the pool bytes of live
    programs are accounted and given back; budgets and the opt-out are honoured.  (Replay is a ``-m gpu`` test.)"""
    import gc

    import quimb_amd.contract as qc
    from quimb_amd.program import ContractionProgram

    arrays, inputs, size = _network(4, 3, "float32")
    shapes = [a.shape for a in arrays]
    xs = [qa.asarray(a) for a in arrays]
    mk = lambda: qa.array_contract_expression(inputs, (), shapes=shapes, optimize="greedy", dtype="float32", cache=False)
    base = qc._PROGRAM_POOL[0]
    expr = mk()
    assert expr._auto_program(xs) is None and expr._auto_program(xs) is None and expr._program is None
    prog = expr._auto_program(xs)
    assert isinstance(prog, ContractionProgram) and expr._auto_program(xs) is prog
    assert prog.inputs == [None] * len(xs)                        # the recorded-on arrays are not kept alive
    with pytest.raises(ValueError):
        prog()
    assert qc._PROGRAM_POOL[0] == base + prog.pool_bytes > base
    del expr, prog
    gc.collect()
    assert qc._PROGRAM_POOL[0] == base
    for kw in (dict(auto_program=False), dict(auto_program_max_bytes=1000), dict(auto_program_total_bytes=1000)):
        with qa.exec_options(**kw):
            expr = mk()             # an expression keeps the options it was BUILT with
        assert [expr._auto_program(xs) for _ in range(4)] == [None] * 4 and expr._program is False
    # host arrays never start one; a tree of fewer than four launches is not worth one
    expr = mk()
    assert [expr._auto_program(arrays) for _ in range(4)] == [None] * 4
    small = qa.array_contract_expression([("a", "b"), ("b", "c")], ("a", "c"), shapes=[(8, 8), (8, 8)], dtype="float32", cache=False)
    two = [qa.asarray(np.ones((8, 8), np.float32))] * 2
    assert [small._auto_program(two) for _ in range(4)] == [None] * 4


def test_aliased_inputs_are_recorded_on_private_copies(recdev):
    """Recording on aliased inputs -- the same array twice (``expr(A, A)``, <psi|psi> with a real ``conj()``), overlapping
    views -- must not bind two inputs to one address range: the later one is recorded on a private copy, so every input
    of the program has a range of its own and a replay on DISTINCT arrays reads each of them (ADVICE round 4, high;
    the replay itself: ``test_program_aliased_recording_replays_on_distinct_inputs`` on the GPU)."""
    inputs = [("a", "b"), ("b", "c"), ("c", "d"), ("d", "e"), ("e", "f")]
    size = {ix: 8 for t in inputs for ix in t}
    ex = qa.TreeExecutor(qa.find_path(inputs, ("a", "f"), size, "greedy"), "float32")
    A = qa.asarray(np.ones((8, 8), np.float32))
    big = qa.asarray(np.ones((2, 8, 8), np.float32))
    view = qa.Array(recdev, big._buf[32:], (8, 8), np.dtype("float32"))       # overlaps the next one half way
    view2 = qa.Array(recdev, big._buf[64:], (8, 8), np.dtype("float32"))
    prog = ex.program([A, A, A, view, view2], strip_exponent=False)
    spans = sorted((x._buf.data_ptr(), x._buf.data_ptr() + x.size * 4) for x in prog.inputs)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), spans
    assert prog.inputs[0] is A and prog.inputs[1] is not A and prog.inputs[2] is not A


def test_pool_reuse_stays_on_the_lane(recdev):
    """A block goes back to the lane that used it last and is handed out again only there; buffers that cross lanes are
    never released."""
    import quimb_amd.program as qp

    log = []
    alloc0, release0 = qp.RecordPool.alloc, qp.RecordPool.release

    def alloc(self, n, tdtype):
        t = alloc0(self, n, tdtype)
        log.append(("a", t.untyped_storage().data_ptr(), self.lane))
        return t

    def release(self, t):
        before = len(self.in_use)
        release0(self, t)
        if len(self.in_use) != before:
            log.append(("r", t.untyped_storage().data_ptr(), self.lane))

    qp.RecordPool.alloc, qp.RecordPool.release = alloc, release
    try:
        arrays, inputs, size = _network(8, 4, "float32")
        ex = qa.TreeExecutor(qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(8, 8)), "float32")
        prog = ex.program([qa.asarray(a) for a in arrays], strip_exponent=True)
    finally:
        qp.RecordPool.alloc, qp.RecordPool.release = alloc0, release0
    owner, reused = {}, 0
    for what, base, lane in log:
        if what == "a":
            if base in owner:
                assert owner[base] == ("free", lane), (base, owner[base], lane)      # handed out again on the SAME lane only
                reused += 1
            owner[base] = ("used", lane)
        else:
            assert owner[base][0] == "used"
            owner[base] = ("free", lane)
    assert reused > 10 and len(prog._pool.blocks) < len([1 for w, *_ in log if w == "a"])


def test_program_refuses_what_it_cannot_record(recdev):
    arrays, inputs, size = _network(4, 3)
    tree = qa.find_slices(qa.ContractionTree(inputs, (), size, path=qa.sweep_path_2d(4, 4)), target_slices=3)
    with pytest.raises(ValueError, match="unsliced"):
        qa.TreeExecutor(tree, "float32").program([qa.asarray(a) for a in arrays])
    ex = qa.TreeExecutor(qa.ContractionTree(inputs, (), size, path=qa.sweep_path_2d(4, 4)), "float32")
    with pytest.raises(ValueError, match="expected 16 arrays"):
        ex.program([qa.asarray(a) for a in arrays[:-1]])
    assert recdev.record is None


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,strip,kind", [("float32", True, "quadrant"), ("float32", False, "quadrant"),
                                              ("float64", True, "sweep"), ("complex64", True, "quadrant"),
                                              ("complex128", False, "quadrant")])
def test_program_replay_matches_launch_by_launch(hip, dtype, strip, kind):
    checks.check_program_replay(6, 4 if np.dtype(dtype).kind == "c" else 6, dtype, strip, kind)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_program_aliased_recording_replays_on_distinct_inputs(hip, dtype):
    checks.check_program_aliasing(dtype)


@pytest.mark.gpu
def test_program_full_size_quadrant_tree(hip):
    """The headline network through a launch program: the fp64 oracle's value at 1e-6, the two joins on gemmk, the
    timing marks deliver their durations, ~100 launches with one host call."""
    import json
    import os

    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_size_oracle.json")))["7"]
    arrays, inputs, size = _network(10, 6)
    arrays, _ = orc.tn2d_rand(10, 10, 6, seed=7, dtype="float32")
    ex = qa.TreeExecutor(qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(10, 10)), "float32")
    prog = ex.program([qa.asarray(a) for a in arrays], strip_exponent=True, mark_min_mults=10**9)
    for slot in range(2):
        m, e = prog(timing_slot=slot)
        m = m.to_numpy().item()
        assert np.sign(m) == ref["sign"] and abs(np.log10(abs(m)) + e - ref["log10_abs"]) < np.log10(1.0 + 1e-6)
    names = [n for (_, _, n, _, _, _) in prog.timings(1)]
    assert sum(n.startswith("gemmk_kernel") for n in names) == 2, names
    assert 20 <= prog.num_launches <= 130      # (94 with every step a launch of its own; 34 with rows 1-4 of every corner fused; 26 with every row one launch)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64", "complex64"])
def test_program_on_general_trees(hip, dtype):
    checks.check_program_on_general_trees(dtype)
