import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, quimb_amd as qa, quimb_amd.device as qd
from quimb_amd.pairwise import plan_rowpass
dev = qd.HipDevice(); qd.set_default_device(dev)
rng = np.random.default_rng(1)
D = 6
ups = [f"v{i}" for i in range(5)]; downs = [f"d{i}" for i in range(5)]; bonds = [f"b{i}" for i in range(4)]
spect = ["s0", "s1", "s2", "s3"]
for kern in ("quad-queue", "quad"):
    sdim = {ix: D for ix in ups + bonds + spect + downs + ["h"]}
    la = tuple(spect + ups)
    sites = [tuple([ups[c]] + ([bonds[c - 1]] if c else []) + [downs[c]] + ([bonds[c]] if c < 4 else ["h"])) for c in range(5)]
    lc = tuple(["h"] + spect + downs)
    rp = plan_rowpass(la, sites, lc, sdim, "float32", kern)
    a = rng.uniform(-0.1, 1, [sdim[i] for i in la]).astype(np.float32)
    ws = [rng.uniform(-0.1, 1, [sdim[i] for i in t]).astype(np.float32) for t in sites]
    xa, xw = qa.asarray(a), [qa.asarray(w) for w in ws]
    out = qa.Array.empty(rp.out_shape, "float32", dev)
    out._buf.fill_(float("nan"))
    dev.contract_rowpass(rp, np.dtype("float32"), xa._buf, [w._buf for w in xw], out._buf, None)
    got = out.to_numpy().astype(np.float64)
    print(kern, "nan count", int(np.isnan(got).sum()))
    num = {ix: i for i, ix in enumerate(sdim)}
    sub = lambda t: [num[ix] for ix in t]
    bad = 0
    for si in range(0, 1296, 7):
        s_ = np.unravel_index(si, (6, 6, 6, 6))
        want = np.einsum(a[s_].astype(np.float64), sub(ups), *[x for w, t in zip(ws, sites) for x in (w.astype(np.float64), sub(t))], sub(["h"] + downs), optimize=True)
        g = got[(slice(None),) + tuple(s_)]
        err = np.abs(g - want) / np.abs(want).max()
        if err.max() > 1e-5:
            bad += 1
            if bad <= 6:
                w_ = np.argwhere(err > 1e-5)
                print("  S", si, "maxerr %.2e" % err.max(), "nbad", len(w_), "d1 values", sorted(set(w_[:, 1])), "h", sorted(set(w_[:, 0])), "first", w_[0], "last", w_[-1])
    print(kern, "bad S:", bad, "of", len(range(0, 1296, 7)))
