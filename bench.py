#!/usr/bin/env python
"""Headline benchmark: contracted-FLOP/s on a 10x10, bond-dimension-6 PEPS
amplitude (single-layer 10x10 D=6 tensor network, fp32), MI355X.

    python bench.py --gpus N --steps K --warmup W

* N == 1 (BASELINE.json configs[2]): one step = one exact contraction of the whole network, inputs resident in HBM,
  the (mantissa, exponent) pair read back after the timed region.  Tree: the faster of the site-by-site boundary
  sweep (8.72e11 multiplications, HBM-bound) and the four-quadrant tree (9.84e11, two 7776^3 MFMA-bound joins) --
  both run untimed first, ``config`` names the one timed and quotes both multiplication counts.
* N  > 1 (configs[3]): the four-quadrant tree with the cut bonds range-sliced over the ranks (halves of 1 / 2 / 3
  bonds at N = 2 / 4 / 8: rank (i, j) of a P x Q grid contracts block (i, j) of the two joins), nothing exchanged on
  the data path, ONE RCCL all-gather of (mantissa, exponent) pairs at the join; strong scaling: ``value`` = the
  one-rank tree's FLOPs / max-over-ranks time.  (256 slices are not reachable with all-6 bonds; ``--sliced`` keeps
  round 1's 216 single-value slices of the sweep, ``--two-sided`` round 2's branch decomposition.)
* N == 1 also reports: ``secondary`` (BASELINE configs #2, #3 as worded -- the compressed boundary-MPS sweep at a stated
  chi, on the headline's tensors -- and #5: circuit amplitude, DMRG2 matvec / local update / sweeps),
  ``scaling_projection`` (the busiest rank's share of the N = 2 / 4 / 8 jobs timed on this one GPU), ``cpu_baseline``, and
  ``split_products_f16x3``: the same contraction re-run with the OPT-IN join arithmetic (Options.join_arith = "f16x3":
  the two joins as three exact fp16 products per multiply-add on the f16 matrix pipe, csrc/gemmh.hip) -- its own ms per
  step, value, error against the fp64 oracle and join timings, BESIDE the headline: ``value`` / ``roofline`` / ``dtype``
  are always the fp32 MFMA path's unless QAMD_JOIN_ARITH=f16x3 makes the split products the process default (``dtype``
  then says so).

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra ``roofline`` (dominant kernel, HIP
events on the launch stream) and ``cpu_baseline`` (numpy/OpenBLAS port of the sweep: a bounded sample to pick the thread
count, then the WHOLE network once on this host, ~70 s) objects.
"""

import argparse
import json
import math
import os
import sys
import time

import numpy as np

# four corner sweeps + the caller's stream want five hardware queues (HIP's default is four: a fifth stream would share
# one and serialise behind it); must be set before the HIP runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# the host driver of this pool only supports dmabuf IPC: without this RCCL (N > 1) fails in hipIpcGetMemHandle
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # spec, /opt/skills/guides/MI355X_MICROARCH.md (6290 measured copy)
MFMA_F32_PEAK_TF = 157.3
MFMA_F16_PEAK_TF = 2516.6   # dense f16 / bf16 MFMA: 256 CUs x 4 SIMDs x 1024 FLOP / cycle x 2.4 GHz (guide: "~2.5 PF dense")


def tn2d_rand(Lx, Ly, D, seed=0, low=-0.1, high=1.0, dtype="float32"):
    """Synthetic 2D network of the reference's accuracy tests (``TN2D_rand`` filled 'mostly positive',
    tests/test_tensor/test_tn2d/test_core.py:243-247), every tensor rescaled by 1 / (mean * D^(legs/2)) so that the
    value stays O(1) in fp32.  Row-major sites; a bond is named ("b", site, site).  The bench's OWN generator (the checker under oracle/ has its own copy; tests/test_oracle.py
    asserts the two produce identical tensors, which is what keys tests/golden/full_size_oracle.json)."""
    rng = np.random.default_rng(seed)
    mean = 0.5 * (low + high)
    bond = lambda a, b: ("b",) + tuple(sorted((a, b)))
    inputs = []
    for r in range(Lx):
        for c in range(Ly):          # per site: left, right, towards row r + 1, towards row r - 1
            t = []
            if c > 0:
                t.append(bond((r, c), (r, c - 1)))
            if c < Ly - 1:
                t.append(bond((r, c), (r, c + 1)))
            if r < Lx - 1:
                t.append(bond((r, c), (r + 1, c)))
            if r > 0:
                t.append(bond((r, c), (r - 1, c)))
            inputs.append(tuple(t))
    arrays = [(rng.uniform(low, high, size=(D,) * len(t)) / (mean * D ** (len(t) / 2.0))).astype(dtype) for t in inputs]
    return arrays, inputs


def build_network(Lx, Ly, D, seed, dtype):
    arrays, inputs = tn2d_rand(Lx, Ly, D, seed=seed, dtype=dtype)
    size = {ix: D for t in inputs for ix in t}
    return arrays, inputs, size


def _result_with_parity(res, args):
    """(mantissa, exponent) of the contraction, plus -- when the fp64 numpy oracle of exactly this full-size network
    has been evaluated (tests/golden/make_full_size_oracle.py, ~10 min on the host, value stored) -- the relative
    error of this run against it (north_star: 1e-6)."""
    out = {"mantissa": res[0], "exponent_log10": res[1]}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "full_size_oracle.json")
    try:
        ref = json.load(open(path)).get(str(args.seed))
    except (OSError, ValueError):
        ref = None
    if ref and (ref["Lx"], ref["Ly"], ref["D"]) == (args.Lx, args.Ly, args.D) and res[0]:
        import math

        log10_abs = math.log10(abs(res[0])) + res[1]
        out["fp64_oracle_log10_abs"] = ref["log10_abs"]
        out["rel_err_vs_fp64_oracle"] = abs(10.0 ** (log10_abs - ref["log10_abs"]) - 1.0)
        out["sign_matches_oracle"] = (res[0] > 0) == (ref["sign"] > 0)
    return out


def cpu_baseline(D, Ly, seed, rows=3, repeats=5, Lx=10, full=True, full_budget_s=240.0):
    """quimb's numpy path restated (oracle/np_oracle.py: per step tensordot through OpenBLAS, strip_exponent) on a
    BOUNDED sample of the same workload: the top ``rows`` rows of the same 10-wide network, same site-by-site
    sweep (rows = 3: one full-size interior row between a first and a last row; the whole network takes ~9 minutes on
    the host: ``full_network_seconds`` quotes the recorded fp64 run of tests/golden/make_full_size_oracle.py).  The BLAS thread count is swept on the 2-row sample first (threadpoolctl; the default of one thread
    per core oversubscribes the skinny 6^9 x 36 x 36 products), then the sample is timed ``repeats`` times after
    one warm-up at the best count and the MEDIAN is reported, with the thread count next to the core count.
    ``full`` (round 6): the WHOLE ``Lx`` x ``Ly`` network once through the same oracle, fp32, same sweep tree, at the swept
    thread count, on THIS host -- ``value`` is then the whole workload's rate and ``full_network_seconds`` is measured beside
    the GPU number, the sample stays as ``sample_value``; skipped (and said so) when the sample predicts more than
    ``full_budget_s`` seconds, so that the default run still ends within minutes."""
    from oracle import np_oracle as orc
    import quimb_amd as qa

    try:
        from threadpoolctl import threadpool_limits
    except ImportError:      # no way to pin: report whatever the BLAS picked
        threadpool_limits = None
    cores = os.cpu_count() or 1

    def sample(nrows):
        arrays, inputs = tn2d_rand(nrows, Ly, D, seed=seed, dtype="float32")
        size = {ix: D for t in inputs for ix in t}
        tree = qa.ContractionTree(inputs, (), size, path=qa.sweep_path_2d(nrows, Ly))
        return arrays, inputs, tree, tree.total_flops("float32")

    def run(smp, threads):
        arrays, inputs, tree, _ = smp
        t0 = time.perf_counter()
        if threadpool_limits is not None and threads:
            with threadpool_limits(limits=threads):
                orc.oracle_array_contract(arrays, inputs, (), path=tree.get_path(), strip_exponent=True)
        else:
            orc.oracle_array_contract(arrays, inputs, (), path=tree.get_path(), strip_exponent=True)
        return time.perf_counter() - t0

    small = sample(2)
    sweep = {}
    for th in sorted({min(cores, t) for t in (16, 32, 64, cores)}):
        sweep[th] = run(small, th)
    best_th = min(sweep, key=sweep.get)
    big = sample(rows)
    run(big, best_th)                                           # warm-up (page faults, BLAS thread pool)
    times = sorted(run(big, best_th) for _ in range(repeats))
    med = times[len(times) // 2]
    out = {
        "value": big[3] / med / 1e12,
        "unit": "TFLOP/s",
        "cores": cores,
        "threads": best_th,
        "kind": "port",
        "sample": f"{rows}x{Ly} D={D} fp32 top-rows sweep of the same network ({big[3]:.3e} of the headline's FLOP), numpy "
                  f"tensordot / OpenBLAS, median of {repeats} after one warm-up: {med:.2f} s (all: "
                  f"{', '.join(f'{t:.2f}' for t in times)})",
        "sample_value": big[3] / med / 1e12,
        "thread_sweep_2row_seconds": {str(k): round(v, 2) for k, v in sweep.items()},
        **_full_network_record(seed),
    }
    if full:
        whole = sample(Lx)
        predicted = whole[3] / (big[3] / med)
        if predicted <= full_budget_s:
            t_full = run(whole, best_th)             # ONE run: the workload itself, not a sample of it
            out.update({
                "value": whole[3] / t_full / 1e12,
                "full_network_seconds": t_full,
                "full_network_flop": whole[3],
                "full_network_note": f"whole {Lx}x{Ly} D={D} network, fp32 numpy oracle (oracle/np_oracle.py), site-by-site sweep "
                                     f"tree, ONE run on this host at {best_th} of {cores} threads; `value` is this run's rate, "
                                     f"`sample_value` the {rows}-row sample's",
            })
            rec = _full_network_record(seed)
            if rec:
                out["full_network_seconds_build_container_fp64"] = rec["full_network_seconds"]
        else:
            out["full_network_skipped"] = f"the sample predicts {predicted:.0f} s > {full_budget_s:.0f} s budget: `value` is the sample's rate"
    return out


def _full_network_record(seed):
    """The anchor of the bounded sample: the WHOLE network through the oracle, recorded once (fp64, same tree)."""
    path = os.path.join(ROOT, "tests", "golden", "full_size_oracle.json")
    try:
        r = json.load(open(path)).get(str(seed))
    except (OSError, ValueError):
        r = None
    if not r or "seconds" not in r:
        return {}
    return {"full_network_seconds": r["seconds"],
            "full_network_note": "whole 10x10 D=6 network, fp64 numpy oracle, same sweep tree, 8 host cores of the build "
                                 "container (tests/golden/make_full_size_oracle.py, recorded with the golden value)"}


def _launch_work(spec):
    """(shape, algorithmic bytes, flops) of ONE launch from its plan record: every operand read once, the result written
    once, no credit for permutes; fp32."""
    if hasattr(spec, "off_k1") and hasattr(spec, "K1"):  # fused pair of steps (Chain2Spec)
        shape = {"fused_steps": 2, "M": spec.M, "D": spec.D, "K": spec.K1, "N": spec.NO * spec.D}
        return shape, 4 * (spec.a_size + spec.c_size + spec.K1 * spec.D**2 + spec.D**3 * spec.NO), 2 * spec.mults
    if hasattr(spec, "w_strides") and hasattr(spec, "s_groups"):  # a whole row of site absorptions (RowpassSpec)
        shape = {"fused_steps": len(spec.sv), "D": spec.D, "out_elems": spec.c_size}
        return shape, 4 * (spec.a_size + spec.c_size + len(spec.sv) * spec.D**4), 2 * spec.mults
    shape = {"B": spec.B, "M": spec.M, "N": spec.N, "K": spec.K}
    return (shape, 4 * spec.B * (spec.M * spec.K + spec.K * spec.N + spec.M * spec.N),
            2 * spec.B * spec.M * spec.N * spec.K)


TINY_BYTES = 4 << 20     # a launch moving less than this is latency, not bandwidth (~12 us on the device)


def roofline_classes(prof_all, step_ms):
    """Per step CLASS of one contraction (one launch-by-launch pass, an HIP event pair around every pairwise launch):
    kernel instantiation, launches, summed device time, and what it reaches of the roof that bounds it --
    min(MFMA peak, arithmetic intensity x HBM peak).  Launches below ``TINY_BYTES`` of algorithmic traffic are one class
    of their own (latency-bound; their event-bracketed times include ~5 us of dispatch each).  The pass runs on ONE
    stream, every launch with the chip to itself: the times add up to the SERIAL step, which the timed region (branches
    on parallel HIP streams) undercuts."""
    cls = {}
    for spec, _, name, _, e0, e1 in prof_all:
        _, nbytes, nflops = _launch_work(spec)
        key = "tiny launches (< 4 MiB of operands + result each)" if nbytes < TINY_BYTES else name
        c = cls.setdefault(key, {"kernel": key, "launches_per_step": 0, "ms_per_step": 0.0, "_b": 0.0, "_f": 0.0, "_names": set()})
        c["launches_per_step"] += 1
        c["ms_per_step"] += e0.elapsed_time(e1)
        c["_b"] += nbytes
        c["_f"] += nflops
        c["_names"].add(name.split("<")[0])
    balance = MFMA_F32_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)
    out = []
    for c in sorted(cls.values(), key=lambda c: -c["ms_per_step"]):
        t = c["ms_per_step"] * 1e-3
        b, f = c.pop("_b"), c.pop("_f")
        names = sorted(c.pop("_names"))
        if c["kernel"].startswith("tiny"):
            c.update(bound="latency", achieved=None, peak=None, unit=None, frac=None, kernels=names,
                     us_per_launch=1e3 * c["ms_per_step"] / c["launches_per_step"])
        elif f / b < balance:
            c.update(bound="hbm", achieved=b / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=b / t / 1e9 / HBM_PEAK_GBS)
        else:
            c.update(bound="mfma", achieved=f / t / 1e12, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=f / t / 1e12 / MFMA_F32_PEAK_TF)
        c["arithmetic_intensity_flop_per_byte"] = f / b
        c["share_of_step_time"] = c["ms_per_step"] / step_ms
        out.append(c)
    return out


def split_products_run(qa, dev, tree, xs, dtype, args, sync, ms_f32, flops_step, shares=None):
    """The same contraction with the OPT-IN join arithmetic (quimb_amd.Options.join_arith = "f16x3", csrc/gemmh.hip): the two
    7776^3 joins as three exact fp16 products per multiply-add on the f16 matrix pipe, fp32 accumulation; everything else
    unchanged.  Reported BESIDE the headline, never as it: `value` / `roofline` above are the fp32 MFMA path's."""
    import time as _t

    try:
        ex16 = qa.TreeExecutor(tree, dtype, options=qa.get_options().replace(join_arith="f16x3"))
        r16 = None
        for _ in range(args.warmup):
            r16 = ex16(xs, strip_exponent=True, defer_exponent=True)
        sync()
        dev.profile_min_mults = 10**9
        dev.profile = []
        try:
            t0 = _t.perf_counter()
            for _ in range(args.steps):
                r16 = ex16(xs, strip_exponent=True, defer_exponent=True)
            sync()
            dt16 = (_t.perf_counter() - t0) / args.steps
        finally:
            prof16, dev.profile = dev.profile, None
        res16 = (r16[0].item(), r16[1] if isinstance(r16[1], float) else dev.read_exponent(r16[1]))
        joins = {}
        for spec, _, name, _, e0, e1 in prof16:
            if not name.startswith("gemmh"):
                continue
            j = joins.setdefault(name, [0.0, 0, _launch_work(spec)])
            j[0] += e0.elapsed_time(e1)
            j[1] += 1
        jl = []
        traffic = None
        try:      # HBM bytes of the product launch from the committed PMC passes (profiles/r06_gemmh_traffic.json)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r06_gemmh_traffic.json")))
            traffic = {"hbm_bytes_per_product_launch": tj["hbm_bytes_per_launch"], "algorithmic_bytes": tj["algorithmic_bytes_per_launch"],
                       "shape": tj["shape"], "source": "profiles/r06_gemmh_traffic.json"}
        except Exception:
            pass
        for name, (tsum, cnt, (shape, _nb, nf)) in sorted(joins.items()):
            avg = tsum / cnt
            jl.append({"kernel": name, "shape": shape, "launches_timed": cnt, "avg_launch_ms": avg,
                       "traffic": traffic if (traffic and traffic["shape"] == shape) else None,
                       "brackets": "the two split passes + the product (one C-ABI call)",
                       "tflops_algorithmic": nf / avg / 1e9, "tflops_executed_on_the_f16_pipe": 3 * nf / avg / 1e9,
                       "frac_of_f16_mfma_peak": 3 * nf / avg / 1e9 / MFMA_F16_PEAK_TF})
        share_ms = None
        if shares is not None:
            # the busiest rank's share of an N-rank job under the same arithmetic (one GPU timing it, as `scaling_projection` does
            # for the fp32 path): what the replicated corner rows leave of the faster joins
            share_ms = {}
            with qa.exec_options(join_arith="f16x3"):
                for w_ in (2, 4, 8):
                    try:
                        share_ms[str(w_)] = shares(w_)
                    except Exception as err:
                        share_ms[str(w_)] = {"error": f"{type(err).__name__}: {err}"}
        return {
            "busiest_rank_share_ms": share_ms,
            "what": "OPT-IN (Options.join_arith = 'f16x3'; default 'f32'): every fp32 operand of the two joins scaled by a power of "
                    "two and split into two fp16 halves (|x - h1 - h2| <= 2^-24 |x|), a b = a1 b1 + a1 b2 + a2 b1 with exact products "
                    "and fp32 accumulation on v_mfma_f32_32x32x16_f16; corner sweeps, exponents, layouts unchanged",
            "ms_per_step": dt16 * 1e3, "steps": args.steps, "warmup": args.warmup,
            "value": flops_step / dt16 / 1e12, "unit": "TFLOP/s (the same algorithmic FLOP count as the headline)",
            "pct_of_fp32_mfma_peak": 100.0 * flops_step / dt16 / 1e12 / MFMA_F32_PEAK_TF,
            "speedup_vs_fp32_mfma_path": ms_f32 / (dt16 * 1e3),
            "result": _result_with_parity(res16, args),
            "joins": jl,
            "accuracy_note": "per entry of a join the error is below an fp32 fma chain's; the f16 instruction's accumulate "
                             "truncates aligned addends ~10 bits below the last place -- a bias of ~2e-7 of an all-positive "
                             "K = 7776 sum, which does not average out in the closing scalar (DESIGN 4.1b)",
        }
    except Exception as err:      # the headline line must survive anything that goes wrong here
        return {"error": f"{type(err).__name__}: {err}"}


def _time_steps(fn, n, sync):
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    sync()
    return (time.perf_counter() - t0) / n, r


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _self_launch(ngpus):
    """``python bench.py --gpus N`` with N > 1 and no launcher environment (WORLD_SIZE unset): start the N ranks
    ourselves -- the same command line under ``python -m torch.distributed.run``, one rank per GPU, rendezvous on
    127.0.0.1 -- and pass the ranks' output through (rank 0 prints the one JSON line).  A launcher that already set
    RANK / WORLD_SIZE (the driver's torchrun) never gets here.  Refuses, loudly, when the box has fewer GPUs than ranks:
    RCCL does not share a device between ranks, and a silent N = 1 run would be recorded as an N-GPU number."""
    import subprocess

    dry = os.environ.get("QAMD_BENCH_DRYRUN") == "1"
    if not dry and os.environ.get("QAMD_BENCH_BACKEND", "nccl") == "nccl":
        import torch

        have = torch.cuda.device_count()
        if have < ngpus:
            print(f"bench.py: --gpus {ngpus} needs {ngpus} visible GPUs, this box has {have} (RCCL does not share a device "
                  f"between ranks; QAMD_BENCH_BACKEND=gloo lets ranks share one GPU as a debugging aid)", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "8")      # (torchrun would set 1 and say so on stderr)
    return subprocess.run(cmd, env=env).returncode


def _dry_run_setup():
    """``QAMD_BENCH_DRYRUN=1``: the script's control flow without a GPU -- the numpy plan interpreter of the test suite
    (tests/emu_device.py) stands in for the device, the few ``torch.cuda`` calls made here are stubbed, ranks talk over
    gloo.  Timings mean nothing; the printed line says ``"dry_run": true`` and can not be mistaken for a measurement.
    Used by tests/test_bench_contract.py to run ``python bench.py --gpus 2`` exactly as a driver without a launcher
    would."""
    import contextlib

    import torch

    class _Stream:
        def wait_stream(self, other):
            pass

    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.device_count = lambda: 1
    torch.cuda.Stream = lambda *a, **k: _Stream()
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import quimb_amd.device as qd
    from emu_device import EmuDevice

    dev = EmuDevice()
    dev.tdev, dev.profile, dev.profile_min_mults = "cpu", None, 0
    qd.set_default_device(dev)
    os.environ["QAMD_BENCH_BACKEND"] = "gloo"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--Lx", type=int, default=10)
    ap.add_argument("--Ly", type=int, default=10)
    ap.add_argument("--D", type=int, default=6)
    ap.add_argument("--slices", type=int, default=216)
    ap.add_argument("--sliced", action="store_true", help="round 1's workload: 216 single-value slices of the sweep (one rank, or round-robin over ranks)")
    ap.add_argument("--two-sided", action="store_true", help="round 2's branch decomposition (top / bottom half sweeps on two rank groups)")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cpu-full", action="store_true", help="cpu_baseline: the bounded 3-row sample only, not the whole network once (~70 s of host time)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs #2 / #5 numbers and the scaling projection (N = 1)")
    ap.add_argument("--tree", choices=["auto", "sweep", "quadrant"], default="auto",
                    help="N = 1 contraction tree: the site-by-site boundary sweep (min-FLOP, HBM-bound), the four-quadrant "
                         "tree (1.13x the multiplications, MFMA-bound joins), or whichever is faster on this device (auto: "
                         "both are run untimed first)")
    ap.add_argument("--inflight", type=int, default=1, help="independent contractions in flight (alternating HIP streams); 1 = one at a time")
    ap.add_argument("--graph", action="store_true", help="N > 1: one hipGraph replay per step instead of launch by launch (measured: no faster -- "
                    "the host enqueues a share in 1.6 ms, the device needs 3.4 -- and the runtime maps the parallel branches of a graph onto fewer queues)")
    ap.add_argument("--launch", choices=["auto", "python", "program"], default="auto",
                    help="how a step is issued: launch by launch from Python, or as a recorded launch program replayed by one "
                         "C call (quimb_amd/program.py).  auto: program for N > 1 (every step ends in a collective, so the "
                         "host's enqueue time is on each step's critical path: 0.33 ms instead of 1.6 ms), Python for N = 1 "
                         "(steps are enqueued back to back and the device never waits for the host; measured equal)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="one GPU times ONE rank's share (the busiest) of a job over this many ranks -- no collective")
    args = ap.parse_args()

    dry_run = os.environ.get("QAMD_BENCH_DRYRUN") == "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.emulate_world:
        sys.exit(_self_launch(args.gpus))
    if dry_run:
        _dry_run_setup()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # QAMD_BENCH_BACKEND=gloo: debugging aid only -- lets the N > 1 code path run with several ranks SHARING one GPU
    # (RCCL refuses duplicate devices); the driver's runs use RCCL, one rank per GPU
    backend = os.environ.get("QAMD_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    import quimb_amd as qa
    from quimb_amd.distributed import contract_sliced, rank_slices
    from quimb_amd.quadrants import QuadrantRank, QuadrantSharding, contract_quadrants

    dev = qa.default_device()
    dtype = "float32"
    arrays, inputs, size = build_network(args.Lx, args.Ly, args.D, args.seed, dtype)
    sweep_tree = qa.ContractionTree(inputs, (), size, path=qa.sweep_path_2d(args.Lx, args.Ly))
    quad_tree = qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(args.Lx, args.Ly)) \
        if min(args.Lx, args.Ly) >= 2 else sweep_tree
    sync = lambda: torch.cuda.synchronize()

    emulate = args.emulate_world if (world == 1 and args.emulate_world > 1) else 0
    if args.sliced:
        mode = "sliced"
    elif args.two_sided:
        mode = "two_sided"
    elif world > 1 or emulate:
        mode = "quadrants"
    else:
        mode = "single"

    tree = quad_tree if args.tree == "quadrant" else sweep_tree
    tree_name = "four quadrants + two joins" if args.tree == "quadrant" else "site-by-site boundary sweep"
    plan = sharding = qrank = None
    tree_probe = None
    xs = [qa.asarray(a) for a in arrays]  # resident in HBM before the timed region
    if mode == "two_sided":
        from quimb_amd.distributed import contract_two_sided, sliced_cols_for_world, two_sided_layout
        from quimb_amd.twosided import TwoSidedContraction

        plan = TwoSidedContraction(inputs, size, args.Lx, args.Ly, dtype, sliced_cols=sliced_cols_for_world([args.D] * args.Ly, world))
    elif mode == "sliced":
        tree = qa.find_slices(sweep_tree, target_slices=args.slices)
        tree_name = "site-by-site boundary sweep"
    elif mode == "quadrants":
        # every rank: the same tree on ITS block of the joins (cut bonds range-sliced); the sharded copies of the site
        # tensors next to the cut are made once, before the timed region (inputs resident in HBM, as at N = 1)
        w_ = emulate or world
        sharding = QuadrantSharding(inputs, size, args.Lx, args.Ly, w_)
        rep0 = sharding.cost_report()
        r_ = int(np.argmax(rep0["per_rank_mults"])) if emulate else rank
        qrank = QuadrantRank(sharding, r_, dtype)
        xs = sharding.shard(xs, r_)
        if args.graph:
            qrank.capture(xs)
        elif args.launch == "program" or (args.launch == "auto" and (world > 1 or emulate)):      # (the emulation runs what a rank runs)
            qrank.program(xs, mark_min_mults=10**9)
        tree, tree_name = quad_tree, "four quadrants + two joins"
    elif args.tree == "auto":
        # both trees, untimed: two warm-up contractions, then the best of three
        tree_probe = {}
        for name, tr in (("site-by-site boundary sweep", sweep_tree), ("four quadrants + two joins", quad_tree)):
            ex_ = qa.TreeExecutor(tr, dtype)
            for _ in range(2):
                ex_(xs, strip_exponent=True)
            best = min(_time_steps(lambda: ex_(xs, strip_exponent=True)[0].item(), 1, sync)[0] for _ in range(3))
            tree_probe[name] = {"ms": best * 1e3, "tree_mults": tr.contraction_cost(),
                                "contraction_width_log2": tr.contraction_width()}
            del ex_
        tree_name = min(tree_probe, key=lambda k: tree_probe[k]["ms"])
        tree = quad_tree if tree_name.startswith("four") else sweep_tree
    ex = qa.TreeExecutor(tree, dtype) if mode in ("single", "sliced") else (qrank.executor if qrank else None)
    single_prog = None
    if mode == "single" and args.launch == "program":
        single_prog = ex.program(xs, strip_exponent=True, mark_min_mults=10**9)
    prog_obj = single_prog or (getattr(qrank, "_program", None) if qrank else None)
    step_no = [0]       # timing slot of a program step inside the timed region
    my = list(rank_slices(tree.nslices, rank, world)) if mode == "sliced" else None

    rank_stats = {}
    timing_on = [False]

    def step():
        if mode == "two_sided":
            return contract_two_sided(plan, xs, strip_exponent=True, stats=rank_stats)
        if mode == "quadrants":
            if getattr(qrank, "_program", None) is not None and timing_on[0] and rank == 0:
                # the marked launches (>= 1e9 multiplications) write their durations into this step's slot
                qrank._program._timing_slot = step_no[0]
                step_no[0] += 1
            if emulate:
                return qrank(xs, defer=True)                     # one rank's share, no collective
            return contract_quadrants(qrank, xs, strip_exponent=True)     # ... + the one all-gather
        if mode == "sliced":
            if world > 1:
                # this rank's slices, then ONE all-reduce of the (mantissa, exponent) pair at the join -- the
                # library routine the gloo tests exercise (tests/test_distributed_gloo.py), RCCL here
                m, e = contract_sliced(ex, xs, strip_exponent=True)
                return float(np.asarray(m).reshape(-1)[0]), e
            m, e = ex(xs, strip_exponent=True, slices=my)
            return m.item(), e
        # unsliced: nothing is read back inside the step -- the (mantissa, exponent) pair stays on the device until the
        # timed region's closing synchronize (steps are enqueued back to back, as a training loop's would be)
        if single_prog is not None:
            slot = step_no[0] if timing_on[0] else None
            step_no[0] += 1
            return single_prog(defer_exponent=True, timing_slot=slot)
        return ex(xs, strip_exponent=True, defer_exponent=True)

    def materialize(r):
        if r is not None and hasattr(r[0], "item"):
            return r[0].item(), (r[1] if isinstance(r[1], float) else dev.read_exponent(r[1]))
        return r

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Launch-by-launch modes keep TWO contractions in flight: consecutive steps go to alternating HIP streams, so the
    # HBM-bound corner sweeps of step i + 1 run under the MFMA-bound joins of step i (the steps are independent
    # contractions; the closing synchronize of the timed region waits for all of them).  --inflight 1 serialises.
    pipelined = args.inflight > 1 and mode in ("single", "quadrants")
    if pipelined:
        rings = [torch.cuda.Stream(device=dev.tdev) for _ in range(args.inflight)]
        for s_ in rings:
            s_.wait_stream(torch.cuda.current_stream(dev.tdev))
        plain_step, counter = step, [0]

        def step():
            counter[0] += 1
            with torch.cuda.stream(rings[counter[0] % len(rings)]):
                return plain_step()

    for _ in range(args.warmup):
        res = step()
    fence()
    # Per-kernel HIP events are recorded inside the timed region (launch-by-launch modes) around the launches that can
    # be the dominant kernel only (>= 1e9 multiplications): an event pair around each of the ~60 tiny first-row
    # launches costs them ~10 us of queue time apiece.  Sliced runs replay one recorded hipGraph per slice, which
    # hides the launches from the host: their kernel timings come from ONE extra, untimed, launch-by-launch pass.
    graphed = mode in ("sliced", "two_sided") or (mode == "quadrants" and args.graph)
    if rank == 0 and not graphed:
        dev.profile_min_mults = 10**9
        dev.profile = []
    if prog_obj is not None:
        # a program carries its own timing events (one slot per step of the timed region; rank 0 reads them); one
        # untimed pass creates them -- on EVERY rank, the steps of a multi-GPU job contain a collective
        timing_on[0] = rank == 0
        for _ in range(args.steps):
            step()
        fence()
        step_no[0] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    dt = time.perf_counter() - t0
    res = materialize(res)
    prof, dev.profile = dev.profile, None
    if prog_obj is not None and rank == 0:
        timing_on[0] = False
        prog_obj._timing_slot = None
        prof = [rec for k in range(args.steps) for rec in prog_obj.timings(k)]
    if graphed:
        # (per-call arguments of the executor, not process state: slices / the share launch by launch for this one pass)
        if rank == 0:
            dev.profile_min_mults = 10**9
            dev.profile = []
        if mode == "quadrants":
            qrank.executor(xs, strip_exponent=True)      # launch by launch, this rank alone (no collective)
        elif mode == "sliced":
            if world > 1:
                contract_sliced(ex, xs, strip_exponent=True, slice_graph=False)
            else:
                ex(xs, strip_exponent=True, slices=my, slice_graph=False)
        else:
            step()
        fence()
        prof, dev.profile = dev.profile, None
    # one more UNTIMED launch-by-launch pass with an event pair around EVERY pairwise launch: the per-class roofline
    # table (``roofline.achieved`` per step class).  Kept out of the timed region because an event pair costs
    # each of the ~60 tiny first-row launches ~10 us of queue time.
    prof_all = None
    if rank == 0 and mode in ("single", "quadrants"):
        # ... and on ONE stream (``lanes=False``, a per-call argument): with the four corner sweeps overlapping, an event pair
        # brackets a launch's share of a contended machine, not the kernel -- the table prices every class with the chip to
        # itself
        dev.profile_min_mults = 0
        dev.profile = []
        try:
            (qrank.executor if mode == "quadrants" else ex)(xs, strip_exponent=True, lanes=False, slice_graph=False)
            torch.cuda.synchronize()
        finally:
            prof_all, dev.profile = dev.profile, None
    scaling_report = None
    if mode == "quadrants":
        # what the ranks executed and how evenly: every rank's own time for its share without the collective
        t_loc, _ = _time_steps(lambda: qrank(xs, defer=True), 3, sync)
        mine_t = torch.tensor([t_loc], dtype=torch.float64, device="cpu" if backend == "gloo" else dev.tdev)
        if world > 1:
            allt = [torch.empty_like(mine_t) for _ in range(world)]
            dist.all_gather(allt, mine_t)
            allt = [float(t.cpu()[0]) for t in allt]
        else:
            allt = [t_loc]
        rep = sharding.cost_report()
        scaling_report = {
            "decomposition": f"four-quadrant tree, cut bonds range-sliced: rank (i, j) of a {rep['grid'][0]} x {rep['grid'][1]} grid "
                             f"contracts block (i, j) of the two joins ({rep['sliced_bonds']} bond(s) in {rep['parts_per_bond']} "
                             f"ranges), no data-path collective, one all-gather of (mantissa, exponent) pairs",
            "flops_useful": 2 * rep["one_rank_mults"], "flops_executed_all_ranks": 2 * rep["executed_mults"],
            "flop_inflation": rep["inflation"], "busiest_rank_fraction_of_one_rank_flops": rep["busiest_rank_fraction"],
            "ideal_speedup_from_flops": rep["ideal_speedup_vs_one_rank"],
            "per_rank_ms_without_collective": [1e3 * t for t in allt],
        }
        if emulate:
            scaling_report["emulated"] = f"ONE GPU timed the busiest rank's share of a {emulate}-rank job; no collective ran"
    elif mode == "two_sided":
        layout = two_sided_layout(plan.nslices, world)
        rep = plan.cost_report(layout)
        scaling_report = {
            "decomposition": f"two-sided: top / bottom half sweeps on {world // 2 or 1} + {world - world // 2} ranks, "
                             f"{plan.nslices} cut-row slice(s) per group, point-to-point hand-off, one all-gather",
            "flops_useful": 2 * rep["useful_mults"], "flops_executed_all_ranks": 2 * rep["executed_mults"],
            "ideal_speedup_from_flops": rep["ideal_speedup_vs_one_rank"],
        }
    tt = torch.tensor([dt], dtype=torch.float64, device=dev.tdev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.cpu()[0])

    if rank == 0:
        # whole job: FLOPs of the tree executed (all slices, hoisted steps once); for the sharded decompositions the
        # USEFUL count -- the one-rank tree's -- so that values at different N compare as time to solution
        if mode == "two_sided":
            flops_step = 2 * plan.one_sided_mults
        elif mode == "quadrants":
            flops_step = 2 * quad_tree.contraction_cost()
        else:
            flops_step = ex.flops()
        ms = dt / args.steps * 1e3
        value = flops_step / (dt / args.steps) / 1e12
        # ---- dominant kernel from HIP-event timings over the timed region ------
        agg = {}
        for spec, dt_, name, sk, e0, e1 in prof:
            shape, nbytes, nflops = _launch_work(spec)
            # (the second join carries the closing inner product in its epilogue -- "<name> + dot", the DOT instantiation of
            # the same kernel on the same GEMM: one class with the plain launches)
            key = (name[:-len(" + dot")] if name.endswith(" + dot") else name, tuple(sorted(shape.items())))
            a = agg.setdefault(key, [0.0, 0, shape, nbytes, nflops, 0])
            a[0] += e0.elapsed_time(e1) * 1e-3
            a[1] += 1
            a[5] += name.endswith(" + dot")
        if os.environ.get("QAMD_BENCH_KERNELS"):   # per-kernel table of the timed region (stderr)
            for (name, shp), (tsum, cnt, _, nb, nf, _nd) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                print(f"  {tsum / args.steps * 1e3:8.3f} ms/step  {cnt // args.steps:4d} x {tsum / cnt * 1e3:7.3f} ms  "
                      f"{nb / (tsum / cnt) / 1e9:7.0f} GB/s {nf / (tsum / cnt) / 1e12:6.1f} TF  {name}  {dict(shp)}", file=sys.stderr)
        roof = None
        if agg:
            key, (tsum, cnt, shape, bytes_launch, flops_launch, ndot) = max(agg.items(), key=lambda kv: kv[1][0])
            cfg = key[0]
            avg = tsum / cnt
            ai = flops_launch / bytes_launch
            # the roof that bounds this launch: min(MFMA peak, AI x HBM peak)
            if ai < MFMA_F32_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9):
                roof = {"bound": "hbm", "achieved": bytes_launch / avg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
            elif cfg.startswith("gemmh"):
                # split products (opt-in): three f16 MFMA products per algorithmic multiply-add; the event pair brackets the
                # two split passes + the product
                roof = {"bound": "mfma", "achieved": 3 * flops_launch / avg / 1e12, "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                        "note": "f16 matrix pipe: achieved = 3 x algorithmic FLOP / (split passes + product); "
                                "algorithmic rate in `tflops`"}
            else:
                roof = {"bound": "mfma", "achieved": flops_launch / avg / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s"}
            roof["frac"] = roof["achieved"] / roof["peak"]
            roof["traffic"] = None
            roof["kernel"] = cfg  # instantiation name from qamd_pair_describe
            # HBM traffic of this kernel from the committed rocprofv3 PMC passes (profiles/):
            # FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, per launch
            import glob

            for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
                try:
                    tj = json.load(open(tpath))
                except Exception:
                    continue
                if tj.get("kernel") == cfg and tj.get("shape") == shape:   # newest round that profiled THIS kernel
                    roof["traffic"] = tj["hbm_bytes_per_launch"]
                    roof["traffic_source"] = "profiles/" + os.path.basename(tpath)
                    break
            roof["shape"] = shape
            roof["arithmetic_intensity_flop_per_byte"] = ai
            roof["tflops"] = flops_launch / avg / 1e12
            roof["avg_launch_ms"] = avg * 1e3
            roof["launches_timed"] = cnt
            if ndot:
                roof["launches_with_fused_inner_product"] = ndot     # rocprofv3 lists them as the <..., true> instantiation
            roof["timed_in"] = ("one untimed launch-by-launch pass after the timed region (the timed region replays hipGraphs)"
                                if graphed else "the timed region (HIP events on the launch stream, launches >= 1e9 multiplications)")
            roof["share_of_step_time"] = tsum / (dt / args.steps * (1 if graphed else args.steps))
            roof["algorithmic_bytes_per_launch"] = bytes_launch
            roof["flops_per_launch"] = flops_launch
        if roof is not None and prof_all:
            roof["classes"] = roofline_classes(prof_all, ms)
            roof["classes_timed_in"] = ("one untimed launch-by-launch pass after the timed region on ONE stream (every launch with "
                                        "the chip to itself), an HIP event pair around every pairwise launch; the timed region "
                                        "overlaps the branches on parallel streams, so the shares can add up to more than 1")
        # ---- N = 1 extras (after the timed region): other BASELINE configs, what one rank of N = 2 / 4 / 8 costs -------
        def share_ms_of(w_, with_report=False):
            """ms per step of the busiest rank's share of a ``w_``-rank job on THIS GPU, under the options in force: what a rank of
            `--gpus N` runs -- its share as a launch program, and a host read after every step (the collective needs the pair): a
            SYNCHRONOUS step, not a pipelined one."""
            sh_ = QuadrantSharding(inputs, size, args.Lx, args.Ly, w_)
            rep_ = sh_.cost_report()
            r_ = int(np.argmax(rep_["per_rank_mults"]))
            qr_ = QuadrantRank(sh_, r_, dtype)
            loc_ = sh_.shard([qa.asarray(a) for a in arrays], r_)
            qr_.program(loc_)

            def one_():
                m_, e_ = qr_(loc_, defer=True)
                return float(e_.cpu()[0]) if hasattr(e_, "cpu") else float(np.asarray(e_).reshape(-1)[0])

            for _ in range(2):
                one_()
            t_, _ = _time_steps(one_, 8, sync)
            del qr_, loc_
            return (t_ * 1e3, rep_) if with_report else t_ * 1e3

        secondary = projection = None
        if mode == "single" and not args.no_secondary and (args.Lx, args.Ly) == (10, 10):
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            try:
                import secondary as _sec

                # (the headline network itself goes along: BASELINE's literal config #3 -- the COMPRESSED boundary-MPS sweep at a
                # stated chi -- is timed on it, its truncation error quoted against the exact value this run just computed)
                par_ = _result_with_parity(res, args)
                secondary = _sec.measure(qa, sync, headline={
                    "arrays": xs, "Lx": args.Lx, "Ly": args.Ly, "D": args.D,
                    "exact_log10_abs_this_run": (math.log10(abs(res[0])) + res[1]) if res and res[0] else None,
                    "fp64_oracle_log10_abs": par_.get("fp64_oracle_log10_abs")})
            except Exception as err:
                secondary = {"error": f"{type(err).__name__}: {err}"}
            projection = {"note": "ONE GPU timing the busiest rank's share of an N-rank job (same code path as --gpus N minus "
                                  "the 16-byte all-gather); the driver's SCALE run is the measurement, this is its preview",
                          "one_gpu_ms": ms}
            for w_ in (2, 4, 8):
                try:
                    t_ms_, rep_ = share_ms_of(w_, True)
                    projection[str(w_)] = {"grid": rep_["grid"], "busiest_rank_ms": t_ms_, "speedup_vs_one_gpu": ms / t_ms_,
                                           "busiest_rank_fraction_of_flops": rep_["busiest_rank_fraction"],
                                           "step": "launch program + host read of the (mantissa, exponent) pair per step"}
                except Exception as err:
                    projection[str(w_)] = {"error": f"{type(err).__name__}: {err}"}
        split16 = None
        if mode == "single" and not args.no_secondary and not dry_run and qa.get_options().join_arith == "f32" \
                and tree is quad_tree:
            split16 = split_products_run(qa, dev, tree, xs, dtype, args, sync, ms, flops_step,
                                         shares=share_ms_of if (args.Lx, args.Ly) == (10, 10) else None)
        cpu = None if (args.no_cpu or world > 1 or emulate) else cpu_baseline(args.D, args.Ly, args.seed, Lx=args.Lx, full=not args.no_cpu_full)   # N=1 only
        nsl = plan.nslices if mode == "two_sided" else tree.nslices
        if mode == "single":
            workload, par = "unsliced, 1 GPU", "single"
        elif mode == "sliced":
            workload, par = f"{tree.nslices} single-value slices over {world} GPU(s), one all-reduce", f"slices{world}"
        elif mode == "two_sided":
            workload, par = f"two-sided branch decomposition over {world} GPU(s), {plan.nslices} cut-row slice(s)", f"branches2xslices{max(world // 2, 1)}"
        else:
            g_ = sharding.cost_report()["grid"]
            workload = (f"BASELINE config #4's role (the same network sharded over sliced indices; 256 slices are not reachable with "
                        f"all-6 bonds): {emulate or world} range-slices of the cut bonds = blocks of a {g_[0]} x {g_[1]} grid over the two "
                        f"joins, one all-gather" + (f" -- EMULATED: one GPU runs the busiest rank's share" if emulate else ""))
            par = f"blocks{g_[0]}x{g_[1]}"
        out = {
            "metric": "contracted-FLOP/s on PEPS amplitude",
            "value": value,
            "unit": "TFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if qa.get_options().join_arith == "f32" else
                     "f32 operands and results; the large joins as f16x3 split products with fp32 accumulation (QAMD_JOIN_ARITH=f16x3: NOT the default)",
            "data": "synthetic" if not dry_run else "synthetic (DRY RUN on the numpy plan interpreter: no GPU, timings void)",
            "config": {
                "workload": f"{args.Lx}x{args.Ly} D={args.D} PEPS amplitude (single-layer TN), exact, " + workload,
                "tree": tree_name,
                "tree_mults": tree.contraction_cost(),
                # the cheapest tree known for this network (quote it next to the executed tree's count, so that
                # an extra-FLOP tree cannot inflate the number): the site sweep on the 10x10 lattice
                "best_known_tree_mults": min(sweep_tree.contraction_cost(), quad_tree.contraction_cost()),
                "trees_tried_untimed": tree_probe,
                "flops_per_step": flops_step,
                "nslices": nsl,
                "contraction_width_log2": tree.contraction_width(),
                "parallelism": par,
                "contractions_in_flight": args.inflight if pipelined else 1,
                "launch": ("launch program: one C call replays the step's recorded launches (quimb_amd/program.py)"
                           if prog_obj is not None else "launch by launch from Python"),
            },
            "pct_mfma_peak": 100.0 * value / (MFMA_F32_PEAK_TF * world),
            # the same time priced by the CHEAPEST known tree's work: what an extra-FLOP tree cannot inflate
            "value_useful_tflops": 2 * min(sweep_tree.contraction_cost(), quad_tree.contraction_cost()) / (dt / args.steps) / 1e12,
            "pct_mfma_peak_useful": 100.0 * 2 * min(sweep_tree.contraction_cost(), quad_tree.contraction_cost())
                                    / (dt / args.steps) / 1e12 / (MFMA_F32_PEAK_TF * world),
            "result": _result_with_parity(res, args) if not emulate else
                      {"mantissa": res[0], "exponent_log10": res[1], "note": "ONE rank's block z_ij of the sum, not the network's value"},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if dry_run:
            out["dry_run"] = True
        if split16 is not None:
            out["split_products_f16x3"] = split16
            if "value" in split16:      # (beside the headline, never as it: `value` above is the fp32 MFMA path's)
                out["value_split_f16x3_opt_in"] = split16["value"]
                out["ms_per_step_split_f16x3_opt_in"] = split16["ms_per_step"]
        if secondary is not None:
            out["secondary"] = secondary
        if projection is not None:
            out["scaling_projection"] = projection
        if scaling_report is not None:
            out["strong_scaling_report"] = scaling_report
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
