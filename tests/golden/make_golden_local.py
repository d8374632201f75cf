#!/usr/bin/env python
"""Golden vectors of the LOCAL operations added in round 2 -- ``Tensor.gate`` (tensor_core.py:3076-3166),
``TensorNetwork.contract_between`` / ``contract_ind`` (:6206-6289) and ``TensorNetwork.trace`` -- produced, like
everything in this directory, by the REAL quimb sources from /root/reference on top of the ``_shims`` stand-ins:

    python tests/golden/make_golden_local.py      ->  tests/golden/local.npz

(kept apart from make_golden.py so that regenerating it does not rewrite the other fixtures)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, "/root/reference")

import quimb.tensor as qtn  # noqa: E402

rng = np.random.default_rng(77)
out = {}
# ---- Tensor.gate: every axis, transpose, preserve_inds=False, complex ------------------------------------------
x = rng.normal(size=(2, 3, 4)) + 1j * rng.normal(size=(2, 3, 4))
t = qtn.Tensor(x, inds=("a", "b", "c"), tags={"T"})
out["gate_x"] = x
cases = []
for ind, d in zip("abc", (2, 3, 4)):
    G = rng.normal(size=(d, d)) + 1j * rng.normal(size=(d, d))
    for transpose in (False, True):
        for preserve in (True, False):
            g = t.gate(G, ind, transpose=transpose, preserve_inds=preserve)
            key = f"gate_{ind}_{int(transpose)}_{int(preserve)}"
            out[key + "_G"] = G
            out[key + "_data"] = np.asarray(g.data)
            cases.append([key, ind, transpose, preserve, list(g.inds)])
out["gate_cases"] = json.dumps(cases)
# ---- contract_between / contract_ind / trace on a ring with one dangling index ------------------------------
shapes = {"i": 3, "j": 4, "k": 2, "l": 3, "o": 5}
spec = [("A", "ij"), ("B", "jk"), ("C", "kl"), ("D", "lio")]
arrs = [rng.normal(size=tuple(shapes[c] for c in inds)) for _, inds in spec]
mk = lambda: qtn.TensorNetwork([qtn.Tensor(a, inds=tuple(i), tags={tg}) for a, (tg, i) in zip(arrs, spec)])
for i, a in enumerate(arrs):
    out[f"ring_{i}"] = a
out["ring_spec"] = json.dumps(spec)
tn = mk()
tn.contract_between("A", "B")
tab = tn["A"]
out["between_inds"] = json.dumps(list(tab.inds))
out["between_tags"] = json.dumps(sorted(tab.tags))
out["between_data"] = np.asarray(tab.data)
out["between_ntensors"] = tn.num_tensors
tn = mk()
tn.contract_ind("l")
tcl = tn["C"]
out["ind_inds"] = json.dumps(list(tcl.inds))
out["ind_tags"] = json.dumps(sorted(tcl.tags))
out["ind_data"] = np.asarray(tcl.data)
out["ring_value"] = np.asarray(mk().contract(all, optimize="greedy").data)
P, Q = rng.normal(size=(3, 4)), rng.normal(size=(4, 3))
op = qtn.TensorNetwork([qtn.Tensor(P, inds=("a", "x")), qtn.Tensor(Q, inds=("x", "b"))])
out["trace_P"], out["trace_Q"] = P, Q
out["trace_value"] = float(op.trace("a", "b"))
np.savez_compressed(os.path.join(HERE, "local.npz"), **out)
print("wrote local.npz:", len(cases), "gate cases; between", out["between_inds"], "; ind", out["ind_inds"], "; trace", out["trace_value"])
