#!/usr/bin/env python
"""Time the fused-pair kernel alone on the headline shape (optionally with ablation bits)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quimb_amd as qa
from quimb_amd.pairwise import plan_chain2
dev = qa.default_device()
D, nm = int(os.environ.get("QAMD_C2_D", "6")), int(os.environ.get("QAMD_C2_NM", "8"))
variant = os.environ.get("QAMD_C2_VARIANT", "interior")   # interior | start (k1 = u only) | end (n2 = one index)
ms = tuple(f"m{i}" for i in range(nm))
if variant == "start":
    la, l1 = ("u", "v") + ms, ("u", "y", "x")
else:
    la, l1 = ("h", "u", "v") + ms, ("h", "y", "u", "x")
lx = ("v", "y") + ms + ("x",)
if variant == "end":
    l2, lc = ("y", "v", "xx"), ms + ("x", "xx")
else:
    l2, lc = ("y", "yy", "v", "xx"), (ms[0], "yy") + ms[1:] + ("x", "xx")
size = {i: D for i in set(la) | set(l1) | set(l2)}
c2 = plan_chain2(la, l1, lx, l2, lc, size, "float32")
assert c2 is not None
# QAMD_C2_DATA: "sym" uniform(-0.5, 0.5) | "pos" uniform(0.1, 1.1) | "const" all ones -- the launch time depends on
# the data through power (DVFS), not through the instruction stream
mode = os.environ.get("QAMD_C2_DATA", "sym")
rnd = {"sym": lambda n: torch.rand(n, device=dev.tdev) - 0.5, "pos": lambda n: torch.rand(n, device=dev.tdev) + 0.1,
       "const": lambda n: torch.ones(n, device=dev.tdev)}[mode]
A = qa.Array(dev, rnd(D ** len(la)), (D,) * len(la), "float32")
W1 = qa.Array(dev, rnd(D ** len(l1)), (D,) * len(l1), "float32")
W2 = qa.Array(dev, rnd(D ** len(l2)), (D,) * len(l2), "float32")
w1p, w2p = W1, W2   # the device addresses the small tensors in place (or packs them itself)
out = qa.Array.empty(c2.out_shape, "float32", dev)
def run():
    dev.contract_chain2(c2, "float32", A._buf, w1p._buf, w2p._buf, out._buf)
run()
torch.cuda.synchronize()
print("kernel:", [e[3] for k, e in dev._pairs.items() if k[0] == "chain2"])
if os.environ.get("QAMD_C2_CHECK", "1") != "0":
    # whole-tensor check against torch (fp32 library GEMMs) -- every element of C
    M = D ** nm
    At = A._buf[: A.size].view((D,) * (len(la) - nm) + (M,))
    w1, w2 = W1._buf[: W1.size].view((D,) * len(l1)), W2._buf[: W2.size].view((D,) * len(l2))
    if variant == "start":
        X = torch.einsum("uvm,uyx->vymx", At, w1)
    else:
        X = torch.einsum("huvm,hyux->vymx", At, w1)
    if variant == "end":
        want = torch.einsum("vymx,yvz->mxz", X, w2)
        got = out._buf[: out.size].view(M, D, D)
    else:
        want = torch.einsum("vymx,ywvz->wmxz", X, w2)           # [yy, m, x, xx]
        got = out._buf[: out.size].view(D, D, M // D, D, D).permute(1, 0, 2, 3, 4).reshape(D, M, D, D)
    del X
    err = ((got - want).abs().max() / want.abs().max()).item()
    bad = int(((got - want).abs() > 1e-4 * want.abs().max()).sum().item())
    if bad and os.environ.get("QAMD_C2_DIAG"):
        idx = ((got - want).abs() > 1e-4 * want.abs().max()).nonzero().cpu().numpy()
        names = ("m", "x", "z") if variant == "end" else ("no", "m", "x", "z")
        for c, n in enumerate(names):
            u = np.unique(idx[:, c] % (64 if n == "m" else 10**9))
            print(f"  bad {n}{' mod 64' if n == 'm' else ''}: {u[:70]}")
        print("  bad m // 64 (chunks):", np.unique(idx[:, names.index('m')] // 64)[:40])
    print(f"check: max |diff| / max |C| = {err:.2e}; elements off by > 1e-4: {bad}; NaN: {int(torch.isnan(got).sum().item())}")
    assert err < 1e-5 and bad == 0
    del want
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
print(f"{variant} data={mode} ablate={os.environ.get('QAMD_CHAIN2_ABLATE','0')} V1={os.environ.get('QAMD_CHAIN2_V1','')} {t*1e3:.3f} ms  {2*c2.mults/t/1e12:.1f} TF  {4*(c2.a_size+c2.c_size)/t/1e9:.0f} GB/s")
# single-launch timing (one event pair per launch) and host-side cost of a launch
import time
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t0 = time.perf_counter()
for _ in range(20): run()
host = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
print(f"single-launch ms: {' '.join(f'{x:.3f}' for x in ts)}   host per launch: {host*1e6:.0f} us")

if os.environ.get("QAMD_TIMING"):
    # library built with -DQAMD_CHAIN2_TIMING: per-wave phase cycle sums come back through absmax_out
    nblk = 4096
    tb = torch.zeros(nblk * 4 * 8, device=dev.tdev)
    dev.contract_chain2(c2, "float32", A._buf, w1p._buf, w2p._buf, out._buf, ep=(None, None, None, tb))
    torch.cuda.synchronize()
    t = tb.cpu().numpy().reshape(-1, 8)
    t = t[t.sum(1) > 0]
    names = (["bookkeep", "s1 side", "s1 rest", "stage 2", "drain", "-", "-", "-"] if "chain2q" in str([e[3] for e in dev._pairs.values() if len(e) > 3])
             else ["s1 t0", "copyout", "ld issue", "s2 t0", "s1 t1", "s2 t1", "s1 t2", "s2 t2"])
    tot = t.sum(1).mean()
    print(f"waves {len(t)}  mean total cycles/wave {tot:.0f}")
    for i, n in enumerate(names):
        print(f"  {n:9s} {t[:, i].mean():10.0f}  {100 * t[:, i].mean() / tot:5.1f}%   (p10 {np.percentile(t[:, i], 10):.0f}  p90 {np.percentile(t[:, i], 90):.0f})")
