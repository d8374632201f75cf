"""fp64 GEMM-shaped contractions of the chi = 512 DMRG matvec (and a few squares) on gemmd.hip against the round-1 kernels:
    python scripts/gemmd_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import quimb_amd as qa

dev = qa.default_device()
rng = np.random.default_rng(0)
chi, w, d = 512, 5, 2
CASES = [
    ("apA,Astb->apstb", dict(a=chi, p=w, A=chi, s=d, t=d, b=chi)),
    ("arstB,brB->astb", dict(a=chi, r=w, s=d, t=d, B=chi, b=chi)),
    ("mk,kn->mn", dict(m=4096, k=4096, n=4096)),
    ("km,kn->mn", dict(m=4096, k=4096, n=4096)),
    ("mk,nk->mn", dict(m=2048, k=2048, n=2048)),
    ("mk,kn->mn", dict(m=1024, k=1024, n=1024)),
]
for eq, dims in CASES:
    lhs, out = eq.split("->")
    ai, bi = lhs.split(",")
    a = qa.asarray(rng.uniform(-0.5, 1.0, [dims[c] for c in ai]))
    b = qa.asarray(rng.uniform(-0.5, 1.0, [dims[c] for c in bi]))
    flop = 2.0 * np.prod([dims[c] for c in set(ai) | set(bi)])
    row = []
    ref = None
    for gd in ("0", "1"):
        dev.force_kernel = 0 if gd == "1" else -2       # plan input -2: automatic choice without the MFMA GEMM kernels
        dev._pairs.clear()
        dev.profile = []
        c = qa.einsum(eq, a, b)
        name = dev.profile[-1][2]
        dev.profile = None
        for _ in range(3):
            qa.einsum(eq, a, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            qa.einsum(eq, a, b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        x = c.to_numpy()
        if ref is None:
            ref = x
        err = float(np.max(np.abs(x - ref)) / np.max(np.abs(ref)))
        row.append(f"{name}: {dt * 1e6:8.1f} us {flop / dt / 1e12:6.1f} TF (diff {err:.1e})")
    print(f"{eq:18s} {str(tuple(dims.values())):40s} | " + " | ".join(row))
dev.force_kernel = 0
