"""bench.py's control flow and its one-JSON-line contract, dry-run on the CPU: the device is the numpy plan interpreter
and the handful of ``torch.cuda`` calls the script makes are stubbed.  Timings are meaningless here -- what is checked
is that every mode runs to the end and prints a line with the keys the driver reads."""
import contextlib
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeStream:
    def wait_stream(self, other):
        pass


@contextlib.contextmanager
def _fake_cuda(monkeypatch):
    import torch

    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(torch.cuda, "Stream", lambda *a, **k: _FakeStream())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _FakeStream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    yield


def _run(monkeypatch, argv):
    sys.path.insert(0, ROOT)
    import bench
    import quimb_amd.device as qd
    from emu_device import EmuDevice

    dev = EmuDevice()
    dev.tdev = "cpu"
    dev.profile = None
    dev.profile_min_mults = 0
    qd.set_default_device(dev)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    out = io.StringIO()
    with _fake_cuda(monkeypatch), contextlib.redirect_stdout(out):
        bench.main()
    lines = [ln for ln in out.getvalue().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.getvalue()
    return json.loads(lines[0])


CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


@pytest.mark.parametrize("argv,tree", [
    (["--Lx", "4", "--Ly", "4", "--D", "3", "--steps", "2", "--warmup", "1", "--no-cpu"], None),
    (["--Lx", "4", "--Ly", "4", "--D", "3", "--steps", "2", "--warmup", "1", "--no-cpu", "--tree", "quadrant"], "four quadrants + two joins"),
    (["--Lx", "4", "--Ly", "4", "--D", "3", "--steps", "2", "--warmup", "1", "--no-cpu", "--tree", "sweep"], "site-by-site boundary sweep"),
    (["--Lx", "4", "--Ly", "4", "--D", "4", "--steps", "2", "--warmup", "1", "--no-cpu", "--emulate-world", "4"], "four quadrants + two joins"),
    (["--Lx", "4", "--Ly", "4", "--D", "3", "--steps", "1", "--warmup", "1", "--no-cpu", "--sliced", "--slices", "9"], "site-by-site boundary sweep"),
    (["--Lx", "4", "--Ly", "4", "--D", "3", "--steps", "1", "--warmup", "1", "--no-cpu", "--inflight", "2"], None),
    (["--Lx", "4", "--Ly", "4", "--D", "3", "--steps", "2", "--warmup", "1", "--no-cpu", "--tree", "quadrant", "--launch", "program"], "four quadrants + two joins"),
])
def test_bench_line_contract(emu, monkeypatch, argv, tree):
    d = _run(monkeypatch, argv)
    assert CONTRACT_KEYS <= set(d), sorted(CONTRACT_KEYS - set(d))
    assert d["metric"] == "contracted-FLOP/s on PEPS amplitude" and d["unit"] == "TFLOP/s" and d["n_gpus"] == 1
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"]
    # the same time priced by the cheapest known tree's work: never above the headline value
    assert 0 < d["value_useful_tflops"] <= d["value"] * (1 + 1e-12) or "slices" in d["config"]["workload"]
    assert d["pct_mfma_peak_useful"] == pytest.approx(100.0 * d["value_useful_tflops"] / 157.3)
    assert d["config"]["best_known_tree_mults"] <= d["config"]["tree_mults"] or "slices" in d["config"]["workload"]
    if tree:
        assert d["config"]["tree"] == tree
    else:
        assert set(d["config"]["trees_tried_untimed"]) == {"site-by-site boundary sweep", "four quadrants + two joins"}
    if "--emulate-world" in argv:
        assert d["strong_scaling_report"]["ideal_speedup_from_flops"] > 1.5 and "EMULATED" in d["config"]["workload"]
    else:
        # the value of the network itself comes back: compare with the oracle on the same generator
        import math

        from oracle import np_oracle as orc

        Lx, Ly, D = (int(argv[argv.index(k) + 1]) for k in ("--Lx", "--Ly", "--D"))
        arrays, inputs = orc.tn2d_rand(Lx, Ly, D, seed=7, dtype="float64")
        want = orc.oracle_array_contract(arrays, inputs, ()).item()
        got = d["result"]["mantissa"] * 10.0 ** d["result"]["exponent_log10"]
        assert got == pytest.approx(want, rel=1e-4)      # (fp32 inputs through the interpreter)


def _bench_worker(rank, world, port, outdir):
    import contextlib as cl

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), QAMD_BENCH_BACKEND="gloo")
    import torch

    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.device_count = lambda: 1
    torch.cuda.Stream = lambda *a, **k: _FakeStream()
    torch.cuda.current_stream = lambda *a, **k: _FakeStream()
    torch.cuda.stream = lambda s: cl.nullcontext()
    import bench
    import quimb_amd.device as qd
    from emu_device import EmuDevice

    dev = EmuDevice()
    dev.tdev, dev.profile, dev.profile_min_mults = "cpu", None, 0
    qd.set_default_device(dev)
    sys.argv = ["bench.py", "--gpus", str(world), "--Lx", "4", "--Ly", "4", "--D", "4", "--steps", "2", "--warmup", "1"]
    out = io.StringIO()
    with cl.redirect_stdout(out):
        bench.main()
    with open(os.path.join(outdir, f"r{rank}.txt"), "w") as f:
        f.write(out.getvalue())


@pytest.mark.parametrize("world", [2, 4])
def test_bench_multi_rank_line(tmp_path, world):
    """``bench.py --gpus N`` as the driver launches it (one process per rank; gloo and stubbed ``torch.cuda`` here):
    rank 0 prints the one line, the others nothing; the value of the network is the oracle's."""
    import socket

    import torch.multiprocessing as mp

    from oracle import np_oracle as orc

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_bench_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [(tmp_path / f"r{r}.txt").read_text() for r in range(world)]
    assert all(not o.strip() for o in outs[1:])
    d = json.loads([ln for ln in outs[0].splitlines() if ln.startswith("{")][0])
    assert CONTRACT_KEYS <= set(d) and d["n_gpus"] == world and d["scaling"] == "strong"
    assert d["config"]["parallelism"].startswith("blocks") and d["cpu_baseline"] is None
    assert d["config"]["launch"].startswith("launch program")      # N > 1: a step is one replayed program + the collective
    assert len(d["strong_scaling_report"]["per_rank_ms_without_collective"]) == world
    arrays, inputs = orc.tn2d_rand(4, 4, 4, seed=7, dtype="float64")
    want = orc.oracle_array_contract(arrays, inputs, ()).item()
    assert d["result"]["mantissa"] * 10.0 ** d["result"]["exponent_log10"] == pytest.approx(want, rel=1e-4)


def test_bench_launches_its_own_ranks():
    """``python bench.py --gpus 2`` with NO launcher environment (a driver that runs exactly that
    got an N = 1 line): the script re-executes itself under ``torch.distributed.run`` with two ranks, rank 0 prints the one
    line and it says ``n_gpus: 2``.  Dry run: plan interpreter + gloo (``QAMD_BENCH_DRYRUN=1``), which the line records."""
    import subprocess

    from oracle import np_oracle as orc

    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "QAMD_BENCH_BACKEND")}
    env["QAMD_BENCH_DRYRUN"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--Lx", "4", "--Ly", "4", "--D", "4",
           "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert CONTRACT_KEYS <= set(d) and d["n_gpus"] == 2 and d["dry_run"] is True and "DRY RUN" in d["data"]
    assert d["config"]["parallelism"] == "blocks2x1" and d["scaling"] == "strong"
    assert len(d["strong_scaling_report"]["per_rank_ms_without_collective"]) == 2
    arrays, inputs = orc.tn2d_rand(4, 4, 4, seed=7, dtype="float64")
    want = orc.oracle_array_contract(arrays, inputs, ()).item()
    assert d["result"]["mantissa"] * 10.0 ** d["result"]["exponent_log10"] == pytest.approx(want, rel=1e-4)


def test_bench_refuses_more_ranks_than_gpus():
    """... and WITHOUT the dry-run switch the self-launcher refuses to start more RCCL ranks than the box has GPUs
    (here: none) instead of recording an N = 1 run as an N-GPU number."""
    import subprocess

    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "QAMD_BENCH_BACKEND", "QAMD_BENCH_DRYRUN")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box could run it")
    assert r.returncode == 2 and "needs 2 visible GPUs" in r.stderr and not r.stdout.strip()


def test_roofline_classes_table():
    """``roofline.classes`` (achieved per step class): launches grouped by kernel instantiation, tiny launches
    lumped as latency-bound, each class priced against min(MFMA peak, AI x HBM peak)."""
    sys.path.insert(0, ROOT)
    import bench
    from quimb_amd.pairwise import GettSpec

    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    def gett(M, N, K):
        return GettSpec(b=(), m=((M, K, 0, N),), n=((N, 0, 1, 1),), k=((K, 1, N, 0),))

    join, stream, tiny = gett(7776, 7776, 7776), gett(6**9, 36, 36), gett(216, 36, 36)
    prof = [(join, None, "gemmk_kernel<3, 4, 2, 2>", 1, Ev(0.0), Ev(7.0)),
            (join, None, "gemmk_kernel<3, 4, 2, 2>", 1, Ev(7.0), Ev(14.0)),
            (stream, None, "sweep_kernel<float>", 1, Ev(0.0), Ev(0.6)),
            (tiny, None, "gett_kernel<float,4,1,4>", 1, Ev(0.0), Ev(0.01)),
            (tiny, None, "stream_kernel<float,2>", 1, Ev(0.0), Ev(0.02))]
    cls = bench.roofline_classes(prof, 16.0)
    assert [c["kernel"].split("<")[0] for c in cls] == ["gemmk_kernel", "sweep_kernel", "tiny launches ("]
    g, sw, t = cls
    assert g["launches_per_step"] == 2 and g["bound"] == "mfma" and g["ms_per_step"] == pytest.approx(14.0)
    assert g["achieved"] == pytest.approx(2 * 7776**3 / 7e-3 / 1e12) and g["frac"] == pytest.approx(g["achieved"] / 157.3)
    assert sw["bound"] == "hbm" and sw["achieved"] == pytest.approx(4 * (6**9 * 72 + 36 * 36) / 0.6e-3 / 1e9)
    assert t["bound"] == "latency" and t["launches_per_step"] == 2 and t["kernels"] == ["gett_kernel", "stream_kernel"]
    assert sum(c["share_of_step_time"] for c in cls) == pytest.approx((14.0 + 0.6 + 0.03) / 16.0)
