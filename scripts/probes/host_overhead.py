"""Host time to ENQUEUE one contraction (no device sync inside the loop) against the device time per step."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import quimb_amd as qa
from bench import build_network
from quimb_amd.quadrants import QuadrantRank, QuadrantSharding
arrays, inputs, size = build_network(10, 10, 6, 7, "float32")
for world in (1, 8):
    sh = QuadrantSharding(inputs, size, 10, 10, world)
    qr = QuadrantRank(sh, world - 1, "float32")
    xs = sh.shard([qa.asarray(a) for a in arrays], world - 1)
    for _ in range(3):
        qr(xs, defer=True)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        qr(xs, defer=True)
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"world {world}: host enqueue {t_host*1e3:.2f} ms/step ({len(qr.executor.plan)} launches -> {t_host/len(qr.executor.plan)*1e6:.1f} us each), wall {t_all*1e3:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): qr(xs, defer=True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
