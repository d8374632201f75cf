"""EXPERIMENT: the two corner sweeps that run BESIDE the first join (their results are only needed by the second) on HIP
streams restricted to K compute units (hipExtStreamCreateWithCUMask): does confining them stop them from slowing the join?
    python scripts/probes/cu_mask.py K [K ...]      (K = 0: ordinary streams)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import quimb_amd as qa
from bench import build_network


def hip():
    path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
    return C.CDLL(path)


def masked_stream(k, ncu=256):
    words = (C.c_uint32 * (ncu // 32))()
    for i in range(k):
        b = (i * (ncu // k + 1)) % ncu if k < ncu else i
        while words[b // 32] >> (b % 32) & 1:
            b = (b + 1) % ncu
        words[b // 32] |= 1 << (b % 32)
    st = C.c_void_p()
    rc = hip().hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(ncu // 32), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


arrays, inputs, size = build_network(10, 10, 6, 7, "float32")
xs = [qa.asarray(a) for a in arrays]
ex = qa.TreeExecutor(qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(10, 10)), "float32")
dev = xs[0]._dev
# lanes whose chains feed only the LAST join
n = len(ex.plan)
prod = ex._producer
kids = [tuple(prod[o] for o in ex._entry_io(e)[0] if o in prod) for e in ex.plan]
anchors = [i for i in range(n) if ex.lanes[i] == 0 and any(ex.lanes[c] != 0 for c in kids[i])]
first = anchors[0]
early = set()
stack = [first]
while stack:
    i = stack.pop()
    if i in early:
        continue
    early.add(i)
    stack.extend(kids[i])
late_lanes = sorted({ex.lanes[i] for i in range(n) if i not in early and ex.lanes[i] != 0})
print("lanes", ex.nlanes, "anchors", anchors, "late lanes", late_lanes, flush=True)


def run(steps=20):
    for _ in range(3):
        ex(xs, strip_exponent=True, defer_exponent=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = ex(xs, strip_exponent=True, defer_exponent=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, r


for k in [int(a) for a in sys.argv[1:]] or [0]:
    dev._lane_pool = {}
    if k:
        for l in late_lanes:
            dev._lane_pool[(l, 0)] = masked_stream(k)
    ms, (m, e) = run()
    print(f"K = {k:3d} CUs for the late corner sweeps: {ms:.3f} ms/step, value {m.to_numpy().item():.6f}e{float(e.cpu()[0]):+.4f}", flush=True)
