#!/usr/bin/env python
"""Headline benchmark: contracted-FLOP/s on a 10x10, bond-dimension-6 PEPS
amplitude (single-layer 10x10 D=6 tensor network, fp32), MI355X.

    python bench.py --gpus N --steps K --warmup W

* N == 1 (BASELINE.json configs[2]): one step = one exact contraction of the
  whole network (site-by-site boundary sweep tree, 1.745e12 FLOP), inputs
  resident in HBM, result read back as an 8-byte (mantissa, exponent).
* N  > 1 (configs[3]): the same network with 216 slices (three bonds of size 6;
  256 is not reachable with all-6 bonds) sharded round-robin over the ranks, one
  RCCL all-reduce of the scalar at the join; strong scaling (total slices fixed).
  ``--sliced`` runs that workload on one GPU too.

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra
``roofline`` (dominant kernel, HIP events on the launch stream) and
``cpu_baseline`` (numpy/OpenBLAS port of the same sweep on a bounded sample)
objects.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # spec, /opt/skills/guides/MI355X_MICROARCH.md (6290 measured copy)
MFMA_F32_PEAK_TF = 157.3


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


def cpu_baseline(D, Ly, seed, rows=4, repeats=3):
    """quimb's numpy path restated (oracle/np_oracle.py: per step tensordot through OpenBLAS, strip_exponent) on a
    BOUNDED sample of the same workload: the top ``rows`` rows of the same 10-wide network, same site-by-site
    sweep (rows = 4: two full-size interior rows, 0.44 of the 1.74 TFLOP; the whole network takes ~9 minutes on the
    host).  The BLAS thread count is swept on the 3-row sample first (threadpoolctl; the default of one thread
    per core oversubscribes the skinny 6^9 x 36 x 36 products), then the sample is timed ``repeats`` times after
    one warm-up at the best count and the MEDIAN is reported, with the thread count next to the core count."""
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

    small = sample(3)
    sweep = {}
    for th in sorted({min(cores, t) for t in (16, 32, 64, cores)}):
        sweep[th] = run(small, th)
    best_th = min(sweep, key=sweep.get)
    big = sample(rows)
    run(big, best_th)                                           # warm-up (page faults, BLAS thread pool)
    times = sorted(run(big, best_th) for _ in range(repeats))
    med = times[len(times) // 2]
    return {
        "value": big[3] / med / 1e12,
        "unit": "TFLOP/s",
        "cores": cores,
        "threads": best_th,
        "kind": "port",
        "sample": f"{rows}x{Ly} D={D} fp32 top-rows sweep of the same network ({big[3]:.3e} of the headline's FLOP), numpy "
                  f"tensordot / OpenBLAS, median of {repeats} after one warm-up: {med:.2f} s (all: "
                  f"{', '.join(f'{t:.2f}' for t in times)})",
        "thread_sweep_3row_seconds": {str(k): round(v, 2) for k, v in sweep.items()},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--Lx", type=int, default=10)
    ap.add_argument("--Ly", type=int, default=10)
    ap.add_argument("--D", type=int, default=6)
    ap.add_argument("--slices", type=int, default=216)
    ap.add_argument("--sliced", action="store_true", help="run the 216-slice workload of round 1 (one rank, or round-robin over ranks)")
    ap.add_argument("--two-sided", action="store_true", help="run the branch decomposition (N > 1 default) on one GPU too")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--tree", choices=["auto", "sweep", "quadrant"], default="auto",
                    help="N = 1 contraction tree: the site-by-site boundary sweep (min-FLOP, HBM-bound), the four-quadrant "
                         "tree (1.13x the multiplications, MFMA-bound joins), or whichever is faster on this device (auto: "
                         "both are run untimed first)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

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

    dev = qa.default_device()
    dtype = "float32"
    arrays, inputs, size = build_network(args.Lx, args.Ly, args.D, args.seed, dtype)
    sweep_tree = qa.ContractionTree(inputs, (), size, path=qa.sweep_path_2d(args.Lx, args.Ly))
    quad_tree = qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(args.Lx, args.Ly)) \
        if min(args.Lx, args.Ly) >= 2 else sweep_tree
    tree = quad_tree if args.tree == "quadrant" else sweep_tree
    tree_name = "four quadrants + two joins" if args.tree == "quadrant" else "site-by-site boundary sweep"
    # N > 1: the BRANCH decomposition (top / bottom half sweeps on two groups of ranks, cut-row slices inside a
    # group, quimb_amd/twosided.py) -- round 1's 216 slices of the one-sided sweep cost 140x the FLOPs
    two_sided = (world > 1 and not args.sliced) or args.two_sided
    sliced = args.sliced and not two_sided
    plan = None
    if two_sided:
        from quimb_amd.distributed import contract_two_sided, sliced_cols_for_world, two_sided_layout
        from quimb_amd.twosided import TwoSidedContraction

        k = sliced_cols_for_world([args.D] * args.Ly, world)
        plan = TwoSidedContraction(inputs, size, args.Lx, args.Ly, dtype, sliced_cols=k)
    if sliced:
        tree = qa.find_slices(tree, target_slices=args.slices)
    xs = [qa.asarray(a) for a in arrays]  # resident in HBM before the timed region
    tree_probe = None
    if args.tree == "auto" and not two_sided and not sliced:
        # both trees, untimed: two warm-up contractions, then the best of three
        tree_probe = {}
        for name, tr in (("site-by-site boundary sweep", sweep_tree), ("four quadrants + two joins", quad_tree)):
            ex_ = qa.TreeExecutor(tr, dtype)
            for _ in range(2):
                ex_(xs, strip_exponent=True)
            best = None
            for _ in range(3):
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                ex_(xs, strip_exponent=True)[0].item()
                torch.cuda.synchronize()
                t_ = time.perf_counter() - t_
                best = t_ if best is None else min(best, t_)
            tree_probe[name] = {"ms": best * 1e3, "tree_mults": tr.contraction_cost(),
                                "contraction_width_log2": tr.contraction_width()}
            del ex_
        tree_name = min(tree_probe, key=lambda k: tree_probe[k]["ms"])
        tree = quad_tree if tree_name.startswith("four") else sweep_tree
    ex = qa.TreeExecutor(tree, dtype) if not two_sided else None
    my = list(rank_slices(tree.nslices, rank, world)) if sliced else None

    rank_stats = {}

    def step():
        if two_sided:
            return contract_two_sided(plan, xs, strip_exponent=True, stats=rank_stats)
        if sliced:
            if world > 1:
                # this rank's slices, then ONE all-reduce of the (mantissa, exponent) pair at the join -- the
                # library routine the gloo tests exercise (tests/test_distributed_gloo.py), RCCL here
                m, e = contract_sliced(ex, xs, strip_exponent=True)
                return float(np.asarray(m).reshape(-1)[0]), e
            m, e = ex(xs, strip_exponent=True, slices=my)
            return m.item(), e
        # unsliced: nothing is read back inside the step -- the (mantissa, exponent) pair stays on the device until the
        # timed region's closing synchronize (steps are enqueued back to back, as a training loop's would be)
        return ex(xs, strip_exponent=True, defer_exponent=True)

    def materialize(r):
        if r is not None and hasattr(r[0], "item"):
            return r[0].item(), dev.read_exponent(r[1]) if hasattr(r[1], "cpu") else r[1]
        return r

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    fence()
    # N = 1 (unsliced): per-kernel HIP events are recorded inside the timed region.  Sliced runs replay one
    # recorded hipGraph per slice, which hides the individual launches from the host: their kernel timings
    # come from ONE extra, untimed, launch-by-launch pass after the timed region.
    if rank == 0 and not sliced and not two_sided:
        # HIP events around the launches that can be the dominant kernel only (>= 1e9 multiplications): an event pair
        # around each of the ~60 tiny first-row launches costs them ~10 us of queue time apiece
        dev.profile_min_mults = 10**9
        dev.profile = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    dt = time.perf_counter() - t0
    res = materialize(res)
    prof, dev.profile = dev.profile, None
    if sliced or two_sided:
        os.environ["QAMD_SLICE_GRAPH"] = "0"
        if rank == 0:
            dev.profile = []
        step()
        fence()
        prof, dev.profile = dev.profile, None
        del os.environ["QAMD_SLICE_GRAPH"]
    scaling_report = None
    if two_sided:
        # honest strong-scaling context: what the ranks executed, how evenly, and the one-GPU unsliced time
        layout = two_sided_layout(plan.nslices, world)
        rep = plan.cost_report(layout)
        mine_t = torch.tensor([rank_stats.get("hoist_s", 0.0), rank_stats.get("compute_s", 0.0)], dtype=torch.float64,
                              device="cpu" if backend == "gloo" else dev.tdev)
        if world > 1:
            allt = [torch.empty_like(mine_t) for _ in range(world)]
            dist.all_gather(allt, mine_t)
            allt = [[float(v) for v in t.cpu()] for t in allt]
        else:
            allt = [[float(v) for v in mine_t.cpu()]]
        one_gpu_ms = None
        if rank == 0:
            ex1 = qa.TreeExecutor(sweep_tree, dtype)
            for _ in range(2):
                ex1(xs, strip_exponent=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                ex1(xs, strip_exponent=True)[0].item()
            torch.cuda.synchronize()
            one_gpu_ms = (time.perf_counter() - t1) / 3 * 1e3
        if world > 1:
            dist.barrier()
        scaling_report = {
            "decomposition": f"two-sided: top / bottom half sweeps on {world // 2 or 1} + {world - world // 2} ranks, "
                             f"{plan.nslices} cut-row slice(s) in contiguous blocks per group, point-to-point hand-off of the "
                             f"cut boundary, one all-gather",
            "flops_useful": 2 * rep["useful_mults"], "flops_executed_all_ranks": 2 * rep["executed_mults"],
            "slice_flop_inflation": rep["inflation"],
            "hoisted_fraction_of_executed": rep["hoisted_mults_all_ranks"] / rep["executed_mults"],
            "ideal_speedup_from_flops": rep["ideal_speedup_vs_one_rank"],
            "per_rank_hoist_ms": [1e3 * a for a, _ in allt], "per_rank_compute_ms": [1e3 * b for _, b in allt],
            "unsliced_one_sided_1gpu_ms_on_rank0": one_gpu_ms,
        }
    tt = torch.tensor([dt], dtype=torch.float64, device=dev.tdev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.cpu()[0])

    if rank == 0:
        # whole job: FLOPs of the tree executed (all slices, hoisted steps once); for the branch decomposition the
        # USEFUL count -- the one-sided sweep's -- so that values at different N compare as time to solution
        flops_step = 2 * plan.one_sided_mults if two_sided else ex.flops()
        ms = dt / args.steps * 1e3
        value = flops_step / (dt / args.steps) / 1e12
        # ---- dominant kernel from HIP-event timings over the timed region ------
        agg = {}
        for spec, dt_, name, sk, e0, e1 in prof:
            if hasattr(spec, "off_k1") and hasattr(spec, "K1"):  # fused pair of steps (Chain2Spec)
                shape = {"fused_steps": 2, "M": spec.M, "D": spec.D, "K": spec.K1, "N": spec.NO * spec.D}
                nbytes = 4 * (spec.a_size + spec.c_size + spec.K1 * spec.D**2 + spec.D**3 * spec.NO)
                nflops = 2 * spec.mults
            elif hasattr(spec, "off_k1"):  # fused triple of steps (Chain3Spec)
                shape = {"fused_steps": 3, "M": spec.a_size // spec.D**4, "D": spec.D, "K": spec.D**2, "N": spec.D**2}
                nbytes = 4 * (spec.a_size + spec.c_size + 3 * spec.D**4)
                nflops = 2 * spec.mults
            else:
                shape = {"B": spec.B, "M": spec.M, "N": spec.N, "K": spec.K}
                nbytes = 4 * spec.B * (spec.M * spec.K + spec.K * spec.N + spec.M * spec.N)
                nflops = 2 * spec.B * spec.M * spec.N * spec.K
            key = (name, tuple(sorted(shape.items())))
            a = agg.setdefault(key, [0.0, 0, shape, nbytes, nflops])
            a[0] += e0.elapsed_time(e1) * 1e-3
            a[1] += 1
        if os.environ.get("QAMD_BENCH_KERNELS"):   # per-kernel table of the timed region (stderr)
            for (name, shp), (tsum, cnt, _, nb, nf) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                print(f"  {tsum / args.steps * 1e3:8.3f} ms/step  {cnt // args.steps:4d} x {tsum / cnt * 1e3:7.3f} ms  "
                      f"{nb / (tsum / cnt) / 1e9:7.0f} GB/s {nf / (tsum / cnt) / 1e12:6.1f} TF  {name}  {dict(shp)}", file=sys.stderr)
        roof = None
        if agg:
            key, (tsum, cnt, shape, bytes_launch, flops_launch) = max(agg.items(), key=lambda kv: kv[1][0])
            cfg = key[0]
            avg = tsum / cnt
            ai = flops_launch / bytes_launch
            # the roof that bounds this launch: min(MFMA peak, AI x HBM peak)
            if ai < MFMA_F32_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9):
                roof = {"bound": "hbm", "achieved": bytes_launch / avg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
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
            roof["timed_in"] = "one untimed launch-by-launch pass after the timed region (the timed region replays hipGraphs)" if sliced else ("rank 0's launches of one untimed pass after the timed region" if two_sided else "the timed region")
            roof["share_of_step_time"] = tsum / (dt / args.steps * (1 if (sliced or two_sided) else args.steps))
            roof["algorithmic_bytes_per_launch"] = bytes_launch
            roof["flops_per_launch"] = flops_launch
        cpu = None if (args.no_cpu or world > 1) else cpu_baseline(args.D, args.Ly, args.seed)   # N=1 only
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
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (
                    f"{args.Lx}x{args.Ly} D={args.D} PEPS amplitude (single-layer TN), exact, "
                    + (f"{tree.nslices} slices over {world} GPU(s), one all-reduce" if sliced else
                       (f"two-sided branch decomposition over {world} GPU(s), {plan.nslices} cut-row slice(s)" if two_sided
                        else "unsliced, 1 GPU"))
                ),
                "tree": tree_name,
                "tree_mults": tree.contraction_cost(),
                "best_known_tree_mults": sweep_tree.contraction_cost(),   # the min-FLOP site sweep
                "trees_tried_untimed": tree_probe,
                "flops_per_step": flops_step,
                "nslices": plan.nslices if two_sided else tree.nslices,
                "contraction_width_log2": tree.contraction_width(),
                "parallelism": f"slices{world}" if sliced else (f"branches2xslices{max(world // 2, 1)}" if two_sided else "single"),
            },
            "pct_mfma_peak": 100.0 * value / (MFMA_F32_PEAK_TF * world),
            "result": _result_with_parity(res, args),
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if scaling_report is not None:
            out["strong_scaling_report"] = scaling_report
            if scaling_report["unsliced_one_sided_1gpu_ms_on_rank0"]:
                out["strong_scaling_report"]["time_vs_unsliced_1gpu"] = ms / scaling_report["unsliced_one_sided_1gpu_ms_on_rank0"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
