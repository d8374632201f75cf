"""chi = 512 effective-Hamiltonian matvec: the hipGraph path of TNLinearOperator(graph=True) against the plain expression path,
which replays a launch program from its third call on (quimb_amd/contract.py)."""
import sys, time; import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, torch, quimb_amd as qa, checks
chi=512
tensors, left, right = checks.dmrg_effective_ham(chi, dtype="float64")
for graph in (True, False):
    A = qa.TNLinearOperator(tensors, left, right, optimize="random-greedy", graph=graph)
    v = qa.asarray(np.random.default_rng(1).standard_normal(A.shape[1]))
    for _ in range(5): A @ v
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(50): w = A @ v
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/50
    print("graph" if graph else "expression (auto program)", f"{dt*1e3:.4f} ms per matvec", type(getattr(A._expr(0), "_program", None)).__name__)
    t0=time.perf_counter()
    e, vec = qa.eigh_lanczos(A, k=1, which="SA", v0=v, ncv=12, tol=1e-14, maxiter=12, miniter=12)
    torch.cuda.synchronize(); print("   lanczos 12 matvecs:", round((time.perf_counter()-t0)*1e3,3), "ms", e)
