#!/usr/bin/env python
"""Golden vectors for the GEMM-shaped split drivers, produced by the REAL quimb (same stand-ins as make_golden.py):
``array_split(x, method=...)`` with "qr:cholesky" (quimb/tensor/decomp.py:2359-2420), "cholesky" (:2262-2322),
"svd:rand" (:1689-1868, seeded: numpy's stream, which quimb_amd.linalg reproduces for a given seed) and "rsvd" (:2538,
on an exactly low-rank input, where the answer does not depend on the random stream).

    python tests/golden/make_golden_decomp.py        ->  tests/golden/decomp.npz
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, "/root/reference")

from quimb.tensor import decomp  # noqa: E402


def main():
    warnings.simplefilter("ignore")
    rng = np.random.default_rng(2024)
    out, cases = {}, []

    def add(name, x, method, **kw):
        left, s, right = decomp.array_split(x, method=method, **kw)
        ci = len(cases)
        out[f"x{ci}"] = x
        for tag, a in (("l", left), ("s", s), ("r", right)):
            if a is not None:
                out[f"{tag}{ci}"] = np.asarray(a)
        cases.append({"name": name, "method": method, "kw": kw})
        print(ci, name, method, kw, [None if a is None else np.shape(a) for a in (left, s, right)])

    tall = rng.normal(size=(24, 12))
    wide = rng.normal(size=(10, 28))
    ctall = rng.normal(size=(20, 8)) + 1j * rng.normal(size=(20, 8))
    for absorb in ("right", "lorthog", "rfactor"):
        add("tall", tall, "qr:cholesky", absorb=absorb)
    for absorb in ("left", "rorthog", "lfactor"):
        add("wide", wide, "qr:cholesky", absorb=absorb)
    add("complex tall", ctall, "qr:cholesky", absorb="right")
    add("tall, no shift", tall, "qr:cholesky", absorb="right", shift=False)
    add("tall, shift 1e-3", tall, "qr:cholesky", absorb="right", shift=1e-3)
    pd = tall.T @ tall
    cpd = ctall.conj().T @ ctall
    for absorb in ("both", "lsqrt", "rsqrt"):
        add("pd", pd, "cholesky", absorb=absorb)
    add("pd, no shift", pd, "cholesky", absorb="both", shift=False)
    add("complex pd", cpd, "cholesky", absorb="both")
    # a decaying spectrum: the sketch + two power iterations resolve the leading values
    spec = np.exp(-0.7 * np.arange(16))
    ul, _ = np.linalg.qr(rng.normal(size=(40, 16)))
    vr, _ = np.linalg.qr(rng.normal(size=(32, 16)))
    low = (ul * spec) @ vr.T
    for absorb in (None, "both", "left", "right"):
        add("decaying 40x32", low, "svd:rand", max_bond=6, absorb=absorb, seed=3)
    add("decaying 32x40", low.T.copy(), "svd:rand", max_bond=6, absorb=None, seed=5)
    add("no power iterations", low, "svd:rand", max_bond=8, absorb=None, seed=7, num_iterations=0, oversample=4)
    # (method_reduced="svd:eig" is left out of the fixtures: under the identity ``njit`` stand-in the reference's numba
    # routine hands the singular values back in ascending order; quimb_amd's own test covers that option)
    add("svd basis", low, "svd:rand", max_bond=6, absorb=None, seed=3, method_lorthog="svd")
    cul, _ = np.linalg.qr(rng.normal(size=(40, 16)) + 1j * rng.normal(size=(40, 16)))
    cvr, _ = np.linalg.qr(rng.normal(size=(32, 16)) + 1j * rng.normal(size=(32, 16)))
    clow = (cul * spec) @ cvr.conj().T               # distinct singular values (a cut inside a degenerate pair is arbitrary)
    add("complex", clow, "svd:rand", max_bond=5, absorb=None, seed=11)
    # exactly rank 6: any randomised range finder of 6 columns is exact
    r6 = (ul[:, :6] * spec[:6]) @ vr[:, :6].T
    add("rank 6", r6, "rsvd", max_bond=6, cutoff=0.0, absorb=None)
    add("rank 6, cutoff", r6, "rsvd", max_bond=6, cutoff=1e-2, cutoff_mode="rel", absorb=None)
    out["cases"] = json.dumps(cases)
    np.savez_compressed(os.path.join(HERE, "decomp.npz"), **out)
    print("wrote decomp.npz")


if __name__ == "__main__":
    main()
