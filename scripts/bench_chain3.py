#!/usr/bin/env python
"""Time the fused-triple kernel (chain3.hip) alone on the headline shape and check it on sampled columns.

env: QAMD_C3_NW = 4 | 8 | 12 (waves per workgroup), QAMD_C3_GRID (workgroups), QAMD_C3_NM (m legs, default 7),
QAMD_C3_DATA = sym | pos | const."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quimb_amd as qa
from quimb_amd.pairwise import plan_chain3

dev = qa.default_device()
D = int(os.environ.get("QAMD_C3_D", "6"))
nm = int(os.environ.get("QAMD_C3_NM", "7"))
ms = tuple(f"m{i}" for i in range(nm))
la = ("h", "a", "b", "c") + ms
l1 = ("h", "h1", "a", "x")
lx1 = ("b", "h1", "c") + ms + ("x",)
l2 = ("h1", "h2", "b", "y")
lx2 = ("c", "h2") + ms + ("x", "y")
l3 = ("h2", "h3", "c", "z")
lc = (ms[0], "h3") + ms[1:] + ("x", "y", "z")
size = {i: D for i in set(la) | set(l1) | set(l2) | set(l3)}
c3 = plan_chain3(la, l1, lx1, l2, lx2, l3, lc, size, "float32")
assert c3 is not None
mode = os.environ.get("QAMD_C3_DATA", "sym")
rnd = {"sym": lambda n: torch.rand(n, device=dev.tdev) - 0.5, "pos": lambda n: torch.rand(n, device=dev.tdev) + 0.1,
       "const": lambda n: torch.ones(n, device=dev.tdev)}[mode]
A = qa.Array(dev, rnd(D ** len(la)), (D,) * len(la), "float32")
W1 = qa.Array(dev, rnd(D ** 4), (D,) * 4, "float32")
W2 = qa.Array(dev, rnd(D ** 4), (D,) * 4, "float32")
W3 = qa.Array(dev, rnd(D ** 4), (D,) * 4, "float32")
out = qa.Array.empty(c3.out_shape, "float32", dev)
out._buf.fill_(float("nan"))


def run():
    dev.contract_chain3(c3, "float32", A._buf, W1._buf, W2._buf, W3._buf, out._buf)


run()
torch.cuda.synchronize()
# ---- check on sampled m columns against numpy fp64 ----
M = D ** nm
rng = np.random.default_rng(0)
cols = np.unique(np.concatenate([rng.integers(0, M, 48), [0, 1, 15, 16, 17, M - 1, M - 16, M // 2]]))
At = A._buf[: A.size].view((D,) * 4 + (M,))[..., torch.as_tensor(cols, device=dev.tdev)].cpu().numpy().astype(np.float64)
w1, w2, w3 = (w.to_numpy().astype(np.float64) for w in (W1, W2, W3))
X1 = np.einsum("habcm,hpax->pxbcm", At, w1)
X2 = np.einsum("pxbcm,pqby->qyxcm", X1, w2)
Cw = np.einsum("qyxcm,qrcz->rmxyz", X2, w3)          # [h3, m, x, y, z]
Cg = out._buf[: out.size].view((D, D, D ** (nm - 1), D, D, D))   # [m0, h3, m_rest, x, y, z]
m0, mr = cols // D ** (nm - 1), cols % D ** (nm - 1)
got = Cg[torch.as_tensor(m0, device=dev.tdev), :, torch.as_tensor(mr, device=dev.tdev)].cpu().numpy()   # [col, h3, x, y, z]
want = Cw.transpose(1, 0, 2, 3, 4)
err = np.abs(got - want).max() / np.abs(want).max()
nan = int(torch.isnan(out._buf[: out.size]).sum().item())
print(f"check: max rel err on {len(cols)} columns = {err:.2e}; unwritten outputs (NaN) = {nan}")
assert err < 2e-5 and nan == 0

for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(f"chain3 D={D} nm={nm} data={mode} NW={os.environ.get('QAMD_C3_NW', '8')} grid={os.environ.get('QAMD_C3_GRID', '256')}: "
      f"{t*1e3:.3f} ms  {2*c3.mults/t/1e12:.1f} TF  {4*(c3.a_size+c3.c_size)/t/1e9:.0f} GB/s   single: {' '.join(f'{x:.3f}' for x in ts)}")
