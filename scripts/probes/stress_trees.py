"""Randomised whole-tree stress: random regular networks (sizes, degrees, bond dimensions, open indices,
strip_exponent, slicing) through TreeExecutor / MicroTree against the numpy oracle."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import quimb_amd as qa, checks
from oracle import np_oracle as orc

bad = 0
tested = 0
kinds = {}
for dt in ("float32", "float64", "complex64", "complex128"):
    hi = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    for seed in range(40):
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.integers(4, 15)); deg = int(rng.integers(2, 5)); D = int(rng.integers(2, 6)); n_out = int(rng.integers(0, 4))
        if (n * deg) % 2:
            n += 1
        try:
            arrays, inputs, output = checks.rand_reg_network(n, deg, D, rng, dt, min(n_out, n))
        except Exception:
            continue
        size = {ix: D for t in inputs for ix in t}
        tree = qa.array_contract_tree(inputs, output, shapes=[a.shape for a in arrays], optimize="greedy")
        if tree.contraction_width() > 24:
            continue
        want = orc.oracle_array_contract([a.astype(hi) for a in arrays], inputs, output, path=tree.get_path())
        ref = max(float(np.max(np.abs(want))), 1e-300)
        tol = 50 * checks.RTOL[np.dtype(dt)]
        def rel(x):
            return float(np.max(np.abs(np.asarray(x) - want))) / ref
        errs = {}
        ex = qa.TreeExecutor(tree, dt)
        errs["exec"] = rel(ex(arrays).to_numpy())
        m, e = ex(arrays, strip_exponent=True)
        errs["strip"] = rel(m.to_numpy() * 10.0**e) if np.isfinite(e) else rel(m.to_numpy() * 0)
        try:
            errs["micro"] = rel(qa.MicroTree(tree, dt)(arrays).to_numpy())
        except ValueError:
            pass
        if len(tree.steps) > 2:
            st = qa.find_slices(tree, target_slices=4)
            if st.nslices > 1:
                errs["sliced"] = rel(qa.TreeExecutor(st, dt)(arrays).to_numpy())
        worst = max(errs.values())
        tested += 1
        for k_ in errs: kinds[k_] = kinds.get(k_, 0) + 1
        if not worst <= tol:
            bad += 1
            print("FAIL", dt, seed, dict(n=n, deg=deg, D=D, n_out=n_out), errs, flush=True)
    print(dt, "done", flush=True)
print("tested:", tested, kinds, "failures:", bad)
