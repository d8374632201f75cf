import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import quimb_amd as qa
from bench import build_network
from quimb_amd.quadrants import QuadrantRank, QuadrantSharding
arrays, inputs, size = build_network(10, 10, 6, 7, "float32")
for world in (8, 1):
    sh = QuadrantSharding(inputs, size, 10, 10, world)
    with qa.exec_options(lane_trace=True):       # an executor keeps the options it is built with
        qr = QuadrantRank(sh, world - 1, "float32")
    xs = sh.shard([qa.asarray(a) for a in arrays], world - 1)
    for _ in range(6):
        qr(xs, defer=True)
    e0 = torch.cuda.Event(enable_timing=True)
    qr(xs, defer=True)          # host runs ahead of this one
    e0.record()
    qr(xs, defer=True)
    torch.cuda.synchronize()
    tr = qr.executor.lane_trace
    base = tr[0][1]
    print(f"--- world {world} ({len(qr.executor.plan)} launches, lanes {qr.executor.nlanes})")
    for name, ev in tr:
        print(f"  {base.elapsed_time(ev)*1e3:9.1f} us  {name}")
