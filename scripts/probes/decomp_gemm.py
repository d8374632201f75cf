"""GPU probe: the GEMM-shaped decompositions against the LAPACK ones at config #5's sizes (fp64):
canonisation 1024 x 512 (QR vs Cholesky-QR with refinement) and the split of a 1024 x 1024 two-site tensor with a decaying
spectrum (gesvd / Gram-eigh / sketch), plus the pieces (potrf, trsm, syevd of the reduced size)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import quimb_amd as qa
from quimb_amd import linalg


def timed(fn, reps=8):
    fn(); fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


rng = np.random.default_rng(0)
a = qa.asarray(rng.standard_normal((1024, 512)))
t, (q, r) = timed(lambda: linalg.qr(a))
print(f"qr 1024x512 (geqrf+orgqr)            {t:8.3f} ms  orth {np.abs(q.to_numpy().T @ q.to_numpy() - np.eye(512)).max():.1e}")
for refine in (False, True):
    t, (q, r) = timed(lambda: linalg.qr_via_cholesky(a, refine=refine))
    qn, rn = q.to_numpy(), r.to_numpy()
    print(f"qr_via_cholesky refine={refine!s:5}          {t:8.3f} ms  orth {np.abs(qn.T @ qn - np.eye(512)).max():.1e}  recon {np.abs(qn @ rn - a.to_numpy()).max():.1e}")
g = torch.randn(522, 522, dtype=torch.float64, device="cuda"); g = g @ g.T + 522 * torch.eye(522, dtype=torch.float64, device="cuda")
for n in (512, 522, 1024):
    gg = g[:n, :n].contiguous() if n <= 522 else (lambda x: x @ x.T + n * torch.eye(n, dtype=torch.float64, device="cuda"))(torch.randn(n, n, dtype=torch.float64, device="cuda"))
    t1, _ = timed(lambda: torch.linalg.cholesky_ex(gg))
    t2, _ = timed(lambda: torch.linalg.eigh(gg))
    L = torch.linalg.cholesky(gg)
    b = torch.randn(1024, n, dtype=torch.float64, device="cuda")
    t3, _ = timed(lambda: torch.linalg.solve_triangular(L.T.contiguous(), b, upper=True, left=False))
    print(f"n={n}: potrf {t1:.3f} ms  syevd {t2:.3f} ms  trsm (1024 x n right) {t3:.3f} ms")
# two-site tensor with a DMRG-like spectrum: 1024 x 1024, singular values decaying to 1e-7 by index ~500
u, _ = np.linalg.qr(rng.standard_normal((1024, 1024)))
v, _ = np.linalg.qr(rng.standard_normal((1024, 1024)))
spec = np.exp(-np.arange(1024) / 30.0)
x = qa.asarray((u * spec) @ v.T)
exact = spec
t, (U, S, VH) = timed(lambda: linalg.svd(x), reps=2)
print(f"split gesvd 1024^2                     {t:8.3f} ms")
t, (U, S, VH) = timed(lambda: linalg.svd_via_eig(x))
s = S.to_numpy(); print(f"split svd_via_eig 1024^2               {t:8.3f} ms  kept {len(s)}  max rel err of s > 1e-6: {np.abs(s[:400] / exact[:400] - 1).max():.1e}")
for q_ in (0, 1):
    for stab in (False, True):
        if q_ == 0 and stab:
            continue
        t, (U, S, VH) = timed(lambda: linalg.svd_rand(x, 512, oversample=10, num_iterations=q_, method_lorthog="qr:cholesky", method_reduced="svd:eig", stabilize=stab))
        s = S.to_numpy()
        rec = (U.to_numpy() * s) @ VH.to_numpy()
        best = (u[:, :len(s)] * spec[:len(s)]) @ v[:, :len(s)].T
        print(f"split svd_rand k=512 q={q_} stabilize={stab!s:5} {t:8.3f} ms  kept {len(s)}  s rel err (first 400) {np.abs(s[:300] / exact[:300] - 1).max():.1e}  |rec - best rank-k| {np.abs(rec - best).max():.1e}")

t, (Q, none, B) = timed(lambda: linalg.svd_rand(x, 512, oversample=0, num_iterations=0, method_lorthog="qr:cholesky", factors_only=True, right=True))
rec = Q.to_numpy() @ B.to_numpy()
print(f"split svd_rand k=512 oversample=0 factors only  {t:8.3f} ms  |Q B - x| {np.abs(rec - x.to_numpy()).max():.1e} (sigma_513 = {spec[512]:.1e})  orth {np.abs(Q.to_numpy().T @ Q.to_numpy() - np.eye(512)).max():.1e}")
