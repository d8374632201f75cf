import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import quimb_amd as qa
from quimb_amd.dmrg import DMRG2, mpo_ham_heis
import checks
sys.path.insert(0, 'scripts/probes')
import opcheck, quimb_amd.dmrg as qdm, quimb_amd.linop as qlo, quimb_amd.eigsolve as qes
opcheck.install(qdm)
hc = []
for k, wk in enumerate(mpo_ham_heis(6, dtype="complex128")):
    u = np.diag([1.0, np.exp(0.37j * (k + 1) ** 2)])
    hc.append(np.einsum("ka,...ab,lb->...kl", u, wk, u.conj()))
dense = checks.mpo_to_dense(hc)
wc, vc = np.linalg.eigh(dense)
dmh = DMRG2(hc, bond_dims=[8], cutoffs=1e-12)
ok = dmh.solve(tol=1e-8, max_sweeps=10, verbosity=1)
print(ok, wc[0], dmh.local_energies[-1])
print(abs(np.vdot(vc[:, 0], checks.mps_to_dense(dmh.state))))
opcheck.report()
