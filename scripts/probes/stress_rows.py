"""Randomised stress of qamd_contract_rowpass on rowq.hip: random spectator counts and sizes (items from a handful to several
rounds of the chip), random extents 1..6 of every new leg, random or canonical index orders on every operand, the static and
the queued item distribution, with and without the fused exponent epilogue -- against fp64 numpy, site by site."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import quimb_amd as qa
import checks
from quimb_amd.pairwise import plan_rowpass

dev = qa.default_device()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 120
D = 6
ups = [f"v{i}" for i in range(5)]; downs = [f"d{i}" for i in range(5)]; bonds = [f"b{i}" for i in range(4)]
worst, fails, t0 = 0.0, 0, time.time()
for case in range(ncase):
    nspect = int(rng.integers(0, 5))
    spect = [f"s{i}" for i in range(nspect)]
    big = rng.random() < 0.25                      # a quarter of the cases: more items than one round of the chip
    sdim = {ix: D for ix in ups + bonds}
    for ix in spect:
        sdim[ix] = int(rng.integers(4, 8)) if big else int(rng.integers(1, 5))
    ext = tuple(int(rng.integers(1, 7)) if rng.random() < 0.5 else 6 for _ in range(6))
    sdim.update(dict(zip(downs + ["h"], ext)))
    canon = rng.random() < 0.5
    la = spect + ups if canon else list(rng.permutation(spect + ups))
    sites = [tuple(rng.permutation([ups[c], downs[c]] + ([bonds[c - 1]] if c else []) + ([bonds[c]] if c < 4 else ["h"]))) for c in range(5)]
    lc = tuple(["h"] + spect + downs) if canon else tuple(rng.permutation(spect + downs + ["h"]))
    kern = "quad-queue" if rng.random() < 0.5 else "quad"
    rp = plan_rowpass(tuple(la), sites, lc, sdim, "float32", kern)
    if rp is None:
        print("case", case, "not planned", nspect, ext); fails += 1; continue
    a = checks.rand(rng, [sdim[i] for i in la], "float32")
    ws = [checks.rand(rng, [sdim[i] for i in t], "float32") for t in sites]
    want = checks._row_reference(a, la, ws, sites, lc)
    xa, xw = qa.asarray(a), [qa.asarray(w) for w in ws]
    out = qa.Array.empty(rp.out_shape, "float32", dev)
    out._buf.fill_(float("nan"))
    ep = None
    scale = 1.0
    if rng.random() < 0.5:                         # the fused exponent epilogue: slots hold the operands' absmax
        mk = lambda x: torch.full((64,), float(np.max(np.abs(x))), dtype=torch.float32, device="cuda")
        ep = (mk(a),) + tuple(mk(w) for w in ws) + (torch.zeros(64, dtype=torch.float32, device="cuda"),)
        scale = float(np.max(np.abs(a))) * float(np.prod([np.max(np.abs(w)) for w in ws]))
    dev.contract_rowpass(rp, np.dtype("float32"), xa._buf, [w._buf for w in xw], out._buf, ep)
    got = out.to_numpy().astype(np.float64) * scale
    ref = max(float(np.max(np.abs(want))), 1e-300)
    err = float(np.max(np.abs(got - want))) / ref if np.isfinite(got).all() else float("inf")
    if ep is not None:
        amax = float(ep[6].max().item()) * scale
        if abs(amax - ref) > 1e-5 * ref:
            print("case", case, "absmax slot", amax, "vs", ref); fails += 1
    worst = max(worst, err)
    if not err <= 1e-6:
        print("case", case, "FAILED", err, dict(nspect=nspect, ext=ext, canon=canon, kern=kern, items=int(np.prod([sdim[i] for i in spect])) * ext[0]))
        fails += 1
print(f"tested: {ncase} rows ({time.time() - t0:.0f} s), worst max-norm relative error {worst:.2e}, failures: {fails}")
sys.exit(1 if fails else 0)
