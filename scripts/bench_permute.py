#!/usr/bin/env python
"""Permute kernel alone: correctness against torch on the same device data + achieved HBM GB/s
(2 x itemsize bytes per element)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quimb_amd as qa

dev = qa.default_device()
CASES = [
    ("2D 16384x16384 transpose", (16384, 16384), (1, 0)),
    ("6^11 reverse", (6,) * 11, tuple(reversed(range(11)))),
    ("6^11 swap last two", (6,) * 11, tuple(range(9)) + (10, 9)),
    ("6^11 rotate left", (6,) * 11, tuple(range(1, 11)) + (0,)),
    ("6^11 rotate right", (6,) * 11, (10,) + tuple(range(10))),
    ("6^11 rotate by two", (6,) * 11, tuple(range(2, 11)) + (0, 1)),
    ("[L,36,R] -> [L,R,36]", (6**4, 36, 6**5), (0, 2, 1)),
    ("[L,R,36] -> [L,36,R]", (6**4, 6**5, 36), (0, 2, 1)),
    ("2^28 bit reversal-ish", (2,) * 28, tuple(range(14, 28)) + tuple(range(14))),
    ("fp64 6^10 swap middle", (6,) * 10, (0, 1, 2, 3, 5, 4, 6, 7, 8, 9)),
    ("small 7x5x3", (7, 5, 3), (2, 0, 1)),
]
for name, shape, perm in CASES:
    dt = "float64" if name.startswith("fp64") else "float32"
    n = int(np.prod(shape))
    x = qa.Array(dev, torch.rand(n, device=dev.tdev, dtype=getattr(torch, dt)), shape, dt)
    y = x.transpose(perm)
    want = x._buf[:n].view(shape).permute(perm).contiguous().view(-1)
    ok = bool(torch.equal(y._buf[:n], want))
    for _ in range(2):
        x.transpose(perm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        x.transpose(perm)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    nb = 2 * x._buf.element_size() * n
    print(f"{name:30s} {'bit-exact' if ok else 'MISMATCH':9s} {t*1e3:8.3f} ms {nb/t/1e9:8.0f} GB/s", flush=True)
    assert ok, name
