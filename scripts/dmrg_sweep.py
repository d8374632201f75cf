#!/usr/bin/env python
"""BASELINE config #5 end to end: ``DMRG2(MPO_ham_heis(L), bond_dims=chi)`` sweeps on the device
(quimb_amd.dmrg.DMRG2; reference quimb/tensor/tn1d/dmrg.py).  Prints wall time per sweep and the energy.

    python scripts/dmrg_sweep.py [L] [chi] [sweeps] [split: svd|eig|rand] [sweep sequence, e.g. RL] [canonize: qr|cholesky]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import quimb_amd as qa
from quimb_amd.dmrg import DMRG2, mpo_ham_heis

L = int(sys.argv[1]) if len(sys.argv) > 1 else 100
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nsweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
split = sys.argv[4] if len(sys.argv) > 4 else "eig"
seq = sys.argv[5] if len(sys.argv) > 5 else "R"            # the reference's default sweeps rightwards every time
canon = sys.argv[6] if len(sys.argv) > 6 else "qr"
oversample = int(sys.argv[7]) if len(sys.argv) > 7 else 10   # split=rand: 0 = no decomposition of the reduced factor (static bond)
import quimb_amd.dmrg as qdm

# where a sweep's wall time goes (device sync around every phase: slightly pessimistic)
phase = {}


def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize()
        phase[name] = phase.get(name, 0.0) + time.perf_counter() - t0
        return out
    return w


if os.environ.get("QAMD_DMRG_PHASES", "1") != "0":
    qdm.eigh_lanczos = timed("local eigensolve (Lanczos + matvecs)", qdm.eigh_lanczos)

    class _L:
        svd = staticmethod(timed("split (gesvd)", qa.linalg.svd))
        svd_via_eig = staticmethod(timed("split (Gram eigh)", qa.linalg.svd_via_eig))
        qr = staticmethod(timed("canonize (QR)", qa.linalg.qr))
        svd_rand = staticmethod(timed("split (sketch + reduced eigh)", qa.linalg.svd_rand))
        qr_via_cholesky = staticmethod(timed("canonize (Cholesky QR)", qa.linalg.qr_via_cholesky))
        lq_via_cholesky = staticmethod(timed("canonize (Cholesky QR)", qa.linalg.lq_via_cholesky))
    qdm.linalg = _L
    qdm.DMRG2._grow_left = timed("environment update", qdm.DMRG2._grow_left)
    qdm.DMRG2._grow_right = timed("environment update", qdm.DMRG2._grow_right)
    qdm.TNLinearOperator = timed("operator setup", qdm.TNLinearOperator)

dm = DMRG2(mpo_ham_heis(L), bond_dims=[chi], cutoffs=1e-10, split=split, canonize=canon, split_opts={"oversample": oversample})
print(f"DMRG2 Heisenberg L={L} fp64, max bond {chi}, split={split}" + (f" (oversample {oversample})" if split == "rand" else "")
      + f", canonize={canon}; start bond {dm.max_bond()}")
prev = "0"
for k in range(nsweeps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    direction = seq[k % len(seq)]
    e = dm.sweep(direction, canonize=not (direction + prev in {"LR", "RL"}), max_bond=chi, cutoff=1e-10)
    prev = direction
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"  sweep {k + 1} ({direction}): {dt:7.2f} s   energy {e:.10f}   E/L {e / L:.8f}   max bond {dm.max_bond()}", flush=True)
    if phase:
        print("      " + "; ".join(f"{n} {v:.2f} s" for n, v in sorted(phase.items(), key=lambda kv: -kv[1]))
              + f"; other {dt - sum(phase.values()):.2f} s", flush=True)
        phase.clear()
print("  (Bethe ansatz, L -> inf: E/L = 1/4 - ln 2 = %.8f)" % (0.25 - np.log(2)))
