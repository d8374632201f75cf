"""Row fusion (csrc/rowpass.hip) on the device against the unfused plan and the fp64 oracle: 4x10 / 6x10 / 10x10 D=6 networks."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import quimb_amd as qa
from oracle import np_oracle as orc

for (Lx, Ly) in ((4, 10), (6, 10), (10, 10)):
    arrays, inputs = orc.tn2d_rand(Lx, Ly, 6, seed=3, dtype="float32")
    inputs = [tuple(t) for t in inputs]
    size = {ix: 6 for t in inputs for ix in t}
    tree = qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(Lx, Ly))
    xs = [qa.asarray(a) for a in arrays]
    res = {}
    for fuse in (True, False):
        with qa.exec_options(fuse_rows=fuse):
            ex = qa.TreeExecutor(tree, "float32")
        nrow = sum(1 for e in ex.plan if e[0] == "rowpass")
        for strip in (False, True):
            r = ex(xs, strip_exponent=strip)
            val = (r[0].to_numpy().item(), r[1]) if strip else (r.to_numpy().item(), 0.0)
            res[(fuse, strip)] = val
        for _ in range(3): ex(xs, strip_exponent=True, defer_exponent=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ex(xs, strip_exponent=True, defer_exponent=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"{Lx}x{Ly} fuse_rows={fuse}: {len(ex.plan)} entries ({nrow} rows), {dt*1e3:.3f} ms/step, values {res[(fuse, False)][0]:.9e} | {res[(fuse, True)][0]:.9f}e{res[(fuse, True)][1]:+.6f}")
    a, b = res[(True, False)][0], res[(False, False)][0]
    sa, sb = res[(True, True)], res[(False, True)]
    print(f"   fused vs unfused: rel {abs(a-b)/abs(b):.2e} (plain), {abs(sa[0]*10**(sa[1]-sb[1])-sb[0])/abs(sb[0]):.2e} (stripped)")
