import numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import quimb_amd as qa
rng = np.random.default_rng(0)
for eq, dims in [("km,kn->nm", dict(k=36, m=46656, n=36)), ("km,kn->nm", dict(k=4, m=4096, n=4)), ("hvm,hxvy->xmy", dict(h=6,v=6,m=46656,x=6,y=6))]:
    lhs, out = eq.split("->"); ai, bi = lhs.split(",")
    a = rng.uniform(-0.5, 1, [dims[c] for c in ai]).astype(np.float32)
    b = rng.uniform(-0.5, 1, [dims[c] for c in bi]).astype(np.float32)
    want = np.einsum(eq, a.astype(np.float64), b.astype(np.float64))
    got = qa.einsum(eq, qa.asarray(a), qa.asarray(b)).to_numpy()
    err = np.abs(got - want)
    bad = np.argwhere(err > 1e-3 * np.abs(want).max())
    print(eq, dims, "max err", err.max(), "nbad", len(bad), "of", want.size)
    if len(bad):
        print(" first bad", bad[:5].tolist(), " last bad", bad[-3:].tolist())
        ax = out.index("m")
        ms = np.unique(bad[:, ax]); print(" bad m count", len(ms), "first", ms[:20].tolist(), "chunks", np.unique(ms // 64)[:20].tolist())
