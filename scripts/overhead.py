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
sys.exit(0)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): run()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
