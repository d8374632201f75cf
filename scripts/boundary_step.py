"""Config #3, approximate mode: ``contract_boundary(max_bond=chi)`` of the 10x10 D=6 fp32 network
 -- accuracy against the exact sweep and where the time goes.

    python scripts/boundary_step.py [chi ...]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: F401  (libamdhip64 before the library)

import quimb_amd as qa
from oracle import np_oracle as orc

Lx = Ly = 10
D = 6
chis = [int(v) for v in sys.argv[1:]] or [8, 16, 36]
arrays, inputs = orc.tn2d_rand(Lx, Ly, D, seed=0, dtype="float32")
size = {ix: D for t in inputs for ix in t}
tree = qa.ContractionTree(inputs, (), size, path=qa.sweep_path_2d(Lx, Ly))
xs = [qa.asarray(a) for a in arrays]
ex = qa.TreeExecutor(tree, "float32")
m, e = ex(xs, strip_exponent=True)
exact = np.log10(abs(m.to_numpy().item())) + e
dev = qa.default_device()
print(f"exact sweep: log10|Z| = {exact:.9f}")
for chi in chis:
    for method in ("svd", "eig"):
        best = None
        for rep in range(2):
            dev.synchronize()
            t0 = time.perf_counter()
            mm, ee = qa.contract_boundary_2d(xs, Lx, Ly, max_bond=chi, strip_exponent=True, method=method)
            dev.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rel = abs(10.0 ** (ee - exact) - 1.0)
        print(f"chi={chi:3d} method={method}: {best * 1e3:8.1f} ms   rel. error vs exact {rel:.2e}", flush=True)
if "--cpu" in sys.argv or True:
    chi = chis[0]
    t0 = time.perf_counter()
    mo, eo = orc.oracle_contract_boundary_2d(arrays, Lx, Ly, max_bond=chi)
    dt = time.perf_counter() - t0
    print(f"numpy oracle chi={chi}: {dt * 1e3:.1f} ms, log10|Z| = {eo + np.log10(abs(mo)):.9f}")
