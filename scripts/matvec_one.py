"""The chi = 512 effective-Hamiltonian matvec of BASELINE config #5, a few applications (rocprofv3 target)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import quimb_amd as qa
chi, w, d = 512, 5, 2
rng = np.random.default_rng(23)
r = lambda *s: rng.uniform(-0.5, 1.0, size=s)
L, R, W1, W2 = r(chi, w, chi), r(chi, w, chi), r(w, w, d, d), r(w, w, d, d)
tensors = [(L, ("a", "p", "A")), (W1, ("p", "q", "s1", "S1")), (W2, ("q", "r", "s2", "S2")), (R, ("b", "r", "B"))]
A = qa.TNLinearOperator(tensors, ("a", "s1", "s2", "b"), ("A", "S1", "S2", "B"), optimize="random-greedy", graph=os.environ.get("GRAPH", "0") == "1")
v0 = qa.asarray(np.random.default_rng(1).standard_normal(chi * d * d * chi))
for _ in range(int(os.environ.get("ITERS", "6"))):
    y = A @ v0
torch.cuda.synchronize()
ex = A._expr(0).executor
for e in ex.plan:
    print(e[0], getattr(e[-1], "kind", ""), getattr(getattr(e[-1], "spec", None), "M", ""), getattr(getattr(e[-1], "spec", None), "N", ""), getattr(getattr(e[-1], "spec", None), "K", ""))
