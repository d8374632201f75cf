import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quimb_amd as qa
rng = np.random.default_rng(5)
dev = qa.default_device()
for (K, M, N) in ((512, 384, 640), (512, 512, 512), (1024, 384, 640), (1296, 1296, 216)):
    a = (rng.uniform(-0.1, 1, (K, M)) + 1j * rng.uniform(-0.1, 1, (K, M))).astype(np.complex64)
    b = (rng.uniform(-0.1, 1, (K, N)) + 1j * rng.uniform(-1, 0.1, (K, N))).astype(np.complex64)
    want = a.astype(np.complex128).T @ b.astype(np.complex128)
    dev.profile, dev.profile_min_mults = [], 0
    with qa.exec_options(join_arith="f16x3-all"):
        got = qa.to_numpy(qa.einsum("km,kn->mn", qa.asarray(a), qa.asarray(b)))
    names = [r[2] for r in dev.profile]; specs = [r[0] for r in dev.profile]
    dev.profile = None
    d = got.astype(np.complex128) - want
    mx = np.abs(want).max()
    print((K, M, N), names, [(s.B, s.M, s.N, s.K) for s in specs], "err re", np.abs(d.real).max() / mx, "im", np.abs(d.imag).max() / mx,
          "mean signed re", d.real.mean() / mx, "im", d.imag.mean() / mx, "max|C|", mx)
    # the same as two real problems through the real path
    ar, ai, br, bi = a.real.copy(), a.imag.copy(), b.real.copy(), b.imag.copy()
    with qa.exec_options(join_arith="f16x3-all"):
        rr = qa.to_numpy(qa.einsum("km,kn->mn", qa.asarray(ar), qa.asarray(br))).astype(np.float64)
        ii = qa.to_numpy(qa.einsum("km,kn->mn", qa.asarray(ai), qa.asarray(bi))).astype(np.float64)
    print("   real products separately: err of (ArBr - AiBi) vs fp64:", np.abs((rr - ii) - want.real).max() / mx)
