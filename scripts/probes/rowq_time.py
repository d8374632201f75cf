"""Time qamd_contract_rowpass (rowq.hip) on the LAST row of a corner sweep of the 10x10 D = 6 network through the Python
boundary: plain and with the fused exponent epilogue, random data and the value range a stripped contraction has."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import quimb_amd as qa
from quimb_amd.pairwise import plan_rowpass

dev = qa.default_device()
D = 6
spect = ["s1", "s2", "s3", "s4"]
ups = [f"v{i}" for i in range(5)]
downs = [f"d{i}" for i in range(5)]
bonds = [f"b{i}" for i in range(4)]
for ext in [(6,) * 6, (3, 6, 6, 6, 6, 6), (3, 3, 6, 6, 6, 6)]:
    size = {ix: D for ix in spect + ups + bonds}
    size.update(dict(zip(downs + ["h"], ext)))
    la = tuple(spect + ups)
    sites = [(ups[c],) + ((bonds[c - 1],) if c else ()) + (downs[c],) + ((bonds[c],) if c < 4 else ("h",)) for c in range(5)]
    lc = tuple(["h"] + spect + downs)
    rng = np.random.default_rng(0)
    for kern in ("quad",):
        rp = plan_rowpass(la, sites, lc, size, "float32", kern)
        a = qa.asarray(rng.uniform(-0.1, 1.0, [size[i] for i in la]).astype(np.float32))
        ws = [qa.asarray(rng.uniform(-0.1, 1.0, [size[i] for i in t]).astype(np.float32)) for t in sites]
        out = qa.Array.empty(rp.out_shape, "float32", dev)
        slots = lambda v: torch.full((64,), v, dtype=torch.float32, device="cuda")
        for label, ep in (("plain", None), ("epilogue", (slots(1.0),) + tuple(slots(1.0) for _ in range(5)) + (torch.zeros(64, dtype=torch.float32, device="cuda"),))):
            for _ in range(3):
                dev.contract_rowpass(rp, np.dtype("float32"), a._buf, [w._buf for w in ws], out._buf, ep)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 20
            for _ in range(n):
                dev.contract_rowpass(rp, np.dtype("float32"), a._buf, [w._buf for w in ws], out._buf, ep)
            e1.record()
            torch.cuda.synchronize()
            us = 1000 * e0.elapsed_time(e1) / n
            print(f"ext {ext} {kern} {label}: {us:8.1f} us  ({2 * rp.mults / us / 1e6:.1f} TFLOP/s)", flush=True)
