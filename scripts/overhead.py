#!/usr/bin/env python
"""Dispatch-overhead probe: many tiny steps (circuit amplitude, MPS -> dense)."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import quimb_amd as qa
import checks
from oracle import np_oracle as orc
rng = np.random.default_rng(0)
arrays, inputs, amp = checks.random_circuit_network(16, 10, rng, "complex64")
arrays_c, inputs_c = arrays, inputs
size = {ix: 2 for t in inputs for ix in t}
tree = qa.array_contract_tree(inputs, (), shapes=[a.shape for a in arrays], optimize="greedy")
ex = qa.TreeExecutor(tree, "complex64")
xs = [qa.asarray(a) for a in arrays]
def run():
    return ex(xs).item()
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): r = run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"circuit 16q depth 10: {len(tree.steps)} steps, {dt*1e3:.2f} ms/contract = {dt/len(tree.steps)*1e6:.1f} us/step, width {tree.contraction_width():.1f}, |err| {abs(r-amp):.2e}")
arrays, inputs = orc.mps_rand(20, 8, 2, seed=1, dtype="float64")
out = tuple(("k", i) for i in range(20))
tree = qa.array_contract_tree(inputs, out, shapes=[a.shape for a in arrays], optimize="greedy")
ex2 = qa.TreeExecutor(tree, "float64"); xs2 = [qa.asarray(a) for a in arrays]
for _ in range(3): ex2(xs2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ex2(xs2)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"MPS L=20 chi=8 -> dense 2^20: {len(tree.steps)} steps, {dt*1e3:.2f} ms/contract = {dt/len(tree.steps)*1e6:.1f} us/step")
g = ex.graph(xs)
for _ in range(3): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): out = g.replay()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"circuit, hipGraph replay: {dt*1e3:.3f} ms/contract = {dt/266*1e6:.2f} us/step, |err| {abs(out.item()-amp):.2e}")
g2 = ex2.graph(xs2)
for _ in range(3): g2.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g2.replay()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"MPS, hipGraph replay: {dt*1e3:.3f} ms/contract")
# the same two as launch programs (what a cached expression switches to from its third call on: contract.py)
for name, exx, xx, nst in (("circuit 16q", ex, xs, 266), ("MPS L=20", ex2, xs2, len(tree.steps))):
    pr = exx.program(xx)
    for _ in range(3): pr(xx)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): o = pr(xx)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name}, launch program: {dt*1e3:.3f} ms/contract = {dt/nst*1e6:.2f} us/step, {pr.num_launches} launches, pool {pr.pool_bytes/1e6:.1f} MB")
    del pr
with qa.exec_options(microtree=False):
    expr = qa.array_contract_expression(inputs_c, (), shapes=[a.shape for a in arrays_c], optimize="greedy", dtype="complex64", cache=False)
for _ in range(4): r = expr(*xs)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): r = expr(*xs)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"circuit 16q through array_contract_expression (launch program from the third call on: {type(expr._program).__name__}): {dt*1e3:.3f} ms/contract, |err| {abs(r.item()-amp):.2e}")
# BASELINE config #2: 53-qubit depth-10 brickwork circuit, one amplitude, complex64 -- 895 tiny steps, width 9:
# dispatch-bound, so the numbers are eager (one Python launch per step) vs one hipGraph replay
arrays, inputs, _ = checks.random_circuit_network(53, 10, np.random.default_rng(0), "complex64", dense=False)
tree = qa.array_contract_tree(inputs, (), shapes=[a.shape for a in arrays], optimize="greedy")
ref = orc.oracle_array_contract([a.astype(np.complex128) for a in arrays], inputs, (), path=tree.get_path())
ex3 = qa.TreeExecutor(tree, "complex64"); xs3 = [qa.asarray(a) for a in arrays]
for _ in range(3): r3 = ex3(xs3).item()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): r3 = ex3(xs3).item()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"circuit 53q depth 10 (config #2): {len(tree.steps)} steps, width {tree.contraction_width():.0f}, eager {dt*1e3:.2f} ms/amplitude "
      f"= {dt/len(tree.steps)*1e6:.1f} us/step, rel err vs fp64 oracle {abs(r3-ref)/abs(ref):.2e}")
g3 = ex3.graph(xs3)
for _ in range(3): g3.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): o3 = g3.replay()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"circuit 53q depth 10, hipGraph replay: {dt*1e3:.3f} ms/amplitude = {dt/len(tree.steps)*1e6:.2f} us/step, "
      f"rel err {abs(o3.item()-ref)/abs(ref):.2e}")
mt = qa.MicroTree(tree, "complex64")
bm = mt.bind(xs3)
for _ in range(3): om = bm()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): om = bm()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"circuit 53q depth 10, MicroTree (one launch walks the tree): {dt*1e3:.3f} ms/amplitude, rel err {abs(om.item()-ref)/abs(ref):.2e}")
nb = 256
e0, e1 = qa.asarray(np.array([1, 0], "complex64")), qa.asarray(np.array([0, 1], "complex64"))
rb = np.random.default_rng(5)
bits = rb.integers(0, 2, size=(nb, 53))
sel = {len(xs3) - 53 + q: ((e0, e1), bits[:, q]) for q in range(53)}
for _ in range(2): ob = bm.batch(sel)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): ob = bm.batch(sel)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
ys = list(xs3)
for q in range(53): ys[len(xs3) - 53 + q] = e1 if bits[7, q] else e0
chk = ex3(ys).item()
print(f"circuit 53q depth 10, MicroTree batch of {nb} bitstrings: {dt*1e3:.3f} ms = {dt/nb*1e6:.1f} us/amplitude "
      f"(instance 7 vs TreeExecutor: rel {abs(ob.to_numpy()[7]-chk)/abs(chk):.1e})")
t0 = time.perf_counter()
for _ in range(3): orc.oracle_array_contract(arrays, inputs, (), path=tree.get_path())
print(f"numpy (oracle) on the host, same tree: {(time.perf_counter()-t0)/3*1e3:.2f} ms/amplitude")
sys.exit(0)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): run()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
