"""Workload for a rocprofv3 --kernel-trace timeline: the busiest rank's share of a WORLD-rank job (or the whole network,
WORLD = 1) as a launch program, STEPS runs, each followed by a host read when MODE = sync.
    python scripts/share_timeline.py WORLD [b2b|sync] [STEPS]"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import quimb_amd as qa
from bench import build_network
from quimb_amd.quadrants import QuadrantRank, QuadrantSharding

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else "b2b"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
arrays, inputs, size = build_network(10, 10, 6, 7, "float32")
xs = [qa.asarray(a) for a in arrays]
if world == 1:
    ex, loc = qa.TreeExecutor(qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(10, 10)), "float32"), xs
else:
    sh = QuadrantSharding(inputs, size, 10, 10, world)
    r = int(np.argmax(sh.cost_report()["per_rank_mults"]))
    ex, loc = QuadrantRank(sh, r, "float32").executor, sh.shard(xs, r)
prog = ex.program(loc, strip_exponent=True)
for _ in range(steps):
    m, e = prog(defer_exponent=True)
    if mode == "sync":
        float(e.cpu()[0])
torch.cuda.synchronize()
print("done", world, mode, prog.num_launches)
