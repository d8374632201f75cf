"""How often DMRG2's rand split falls back from CholeskyQR to Householder QR in a chi = 512 fp64 sweep, and what a sweep costs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quimb_amd as qa
from quimb_amd import linalg
from quimb_amd.dmrg import DMRG2, mpo_ham_heis

calls = {"n": 0, "fallback": 0}
orig = linalg.orth_cholesky_checked
def counted(y):
    q, fb = orig(y)
    calls["n"] += 1
    calls["fallback"] += int(fb)
    return q, fb
linalg.orth_cholesky_checked = counted
dev = qa.default_device()
dm = DMRG2(mpo_ham_heis(100), bond_dims=[512], cutoffs=1e-10, split="rand", canonize="cholesky", split_opts={"oversample": 0})
for sw in range(3):
    dev.synchronize(); t0 = time.perf_counter()
    e = dm.sweep("R", canonize=True, max_bond=512, cutoff=1e-10)
    dev.synchronize()
    print(f"sweep {sw}: {time.perf_counter() - t0:.3f} s, E = {float(e):.6f}, orth calls {calls['n']}, Householder fallbacks {calls['fallback']}", flush=True)
