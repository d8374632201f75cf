#!/usr/bin/env python
"""BASELINE config #5 end to end: ``DMRG2(MPO_ham_heis(L), bond_dims=chi)`` sweeps on the device
(quimb_amd.dmrg.DMRG2; reference quimb/tensor/tn1d/dmrg.py).  Prints wall time per sweep and the energy.

    python scripts/dmrg_sweep.py [L] [chi] [sweeps] [split]
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
dm = DMRG2(mpo_ham_heis(L), bond_dims=[chi], cutoffs=1e-10, split=split)
print(f"DMRG2 Heisenberg L={L} fp64, max bond {chi}, split={split}; start bond {dm.max_bond()}")
prev = "0"
for k in range(nsweeps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e = dm.sweep("R", canonize=True, max_bond=chi, cutoff=1e-10)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"  sweep {k + 1}: {dt:7.2f} s   energy {e:.10f}   E/L {e / L:.8f}   max bond {dm.max_bond()}", flush=True)
print("  (Bethe ansatz, L -> inf: E/L = 1/4 - ln 2 = %.8f)" % (0.25 - np.log(2)))
