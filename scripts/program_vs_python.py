"""Step time of the headline contraction (and of one rank's share of an N-rank job) launch by launch from Python against
the same launch sequence replayed as a launch program (quimb_amd/program.py).   python scripts/program_vs_python.py"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import quimb_amd as qa
from bench import build_network
from quimb_amd.quadrants import QuadrantRank, QuadrantSharding


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, t_host * 1e3


def timed_sync(fn, n=20):
    """every step followed by a host read of its result -- what a rank of a multi-GPU job does (the collective needs
    the value), and what a lone contraction costs"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m, e = fn()
        float(e.cpu()[0])
    return (time.perf_counter() - t0) / n * 1e3


arrays, inputs, size = build_network(10, 10, 6, 7, "float32")
xs = [qa.asarray(a) for a in arrays]
for world in (1, 2, 4, 8):
    if world == 1:
        ex = qa.TreeExecutor(qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(10, 10)), "float32")
        loc = xs
    else:
        sh = QuadrantSharding(inputs, size, 10, 10, world)
        r = int(np.argmax(sh.cost_report()["per_rank_mults"]))
        ex = QuadrantRank(sh, r, "float32").executor
        loc = sh.shard(xs, r)
    py = timed(lambda: ex(loc, strip_exponent=True, defer_exponent=True))
    prog = ex.program(loc, strip_exponent=True)
    pr = timed(lambda: prog(defer_exponent=True))
    py_s = timed_sync(lambda: ex(loc, strip_exponent=True, defer_exponent=True))
    pr_s = timed_sync(lambda: prog(defer_exponent=True))
    m, e = prog()
    print(f"world {world}: back to back: python loop {py[0]:.3f} ms/step (host enqueue {py[1]:.3f}) | program {pr[0]:.3f} "
          f"(host enqueue {pr[1]:.3f});  one step then read: python loop {py_s:.3f} | program {pr_s:.3f};  "
          f"{prog.num_launches} launches, pool {prog.pool_bytes / 1e6:.0f} MB; value {m.to_numpy().item():.6f}e{e:+.6f}")
