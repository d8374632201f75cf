#!/usr/bin/env python
"""The REAL quimb (sources under /root/reference, third-party imports answered by ``_shims``) running ON
``quimb_amd`` arrays: the drop-in boundary B1 of SURVEY.md section 8b exercised by quimb's own code.

The ``autoray`` stand-in dispatches exactly like autoray does -- by the top-level module of the array's class --
so every ``do("tensordot" / "einsum" / "transpose" / "reshape" / "linalg.svd" ...)`` quimb issues lands on the
module-level functions of ``quimb_amd``; ``quimb_amd.autoray_backend.register()`` adds the overrides a
maintainer would register (fuse, norm_fro, the split drivers).  The device is the numpy plan interpreter of the
CPU tests (this script runs in the build container only: /root/reference does not exist on the GPU box).

Run by tests/test_dropin_reference.py in a subprocess; prints "DROPIN OK" when every check passed.
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
_ap = argparse.ArgumentParser()
_ap.add_argument("--device", choices=["emu", "hip"], default="emu",
                 help="emu: the numpy plan interpreter of the CPU tests; hip: the MI355X (needs a GPU AND the quimb sources)")
_ap.add_argument("--stack", choices=["shims", "real"], default="shims",
                 help="shims: quimb's sources from /root/reference with this repo's stand-ins for autoray / cotengra / numba / "
                      "cytoolz; real: the INSTALLED quimb, autoray and cotengra (scripts/verify_real_stack.py)")
ARGS = _ap.parse_args()
if ARGS.stack == "shims":
    sys.path.insert(0, os.path.join(HERE, "_shims"))
    sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402
import quimb.tensor as qtn  # noqa: E402

import quimb_amd as qa  # noqa: E402
import quimb_amd.autoray_backend as qab  # noqa: E402
import quimb_amd.device as qd  # noqa: E402

if ARGS.device == "emu":
    from emu_device import EmuDevice  # noqa: E402

    dev = EmuDevice()
    qd.set_default_device(dev)
else:
    dev = qa.default_device()                      # HipDevice: raises if no GPU is visible
    import collections

    dev.calls = collections.Counter()
    for _name in ("contract_pair", "permute"):     # the same launch counters the interpreter keeps
        def _counted(*a, _f=getattr(dev, _name), _n=_name, **k):
            dev.calls[_n] += 1
            return _f(*a, **k)
        setattr(dev, _name, _counted)
assert qab.register() == "quimb_amd"
print(f"stack: {ARGS.stack} (quimb from {os.path.dirname(qtn.__file__)}), device: {type(dev).__name__}")


def host(x):
    return np.asarray(x.to_numpy() if hasattr(x, "to_numpy") else x)


def on_device(tn):
    tn = tn.copy()
    tn.apply_to_arrays(qa.asarray)                                   # tensor_core.py:5304
    assert all(type(t.data) is qa.Array and t.backend == "quimb_amd" for t in tn)
    return tn


def launched():
    return dev.calls["contract_pair"]


# 1. TensorNetwork.contract / strip_exponent on a 2D lattice
tn = qtn.TN2D_rand(4, 4, 3, seed=42, dtype="float64")
want = tn.contract(all, optimize="greedy")
n0 = launched()
got = on_device(tn).contract(all, optimize="greedy")
assert launched() - n0 == 15, "every pairwise step must go through the backend's contraction"
assert type(got) is qa.Array and abs(float(got) / want - 1) < 1e-12
m, e = on_device(tn).contract(all, optimize="greedy", strip_exponent=True)
assert abs(float(m) * 10**e / want - 1) < 1e-12

# 2. Tensor-level layout ops and @ (tensor_core.py:3786, :3252-3373)
t = tn.tensors[5]
td = on_device(tn).tensors[5]
assert np.array_equal(host(td.fuse({"x": t.inds[:2]}).data), t.fuse({"x": t.inds[:2]}).data)
assert np.array_equal(host(td.transpose(*t.inds[::-1]).data), t.transpose(*t.inds[::-1]).data)
assert np.array_equal(host(td.isel({t.inds[0]: 1}).data), t.isel({t.inds[0]: 1}).data)
a, b = tn.tensors[0], tn.tensors[1]
ad, bd = on_device(tn).tensors[:2]
assert np.allclose(host((ad @ bd).data), (a @ b).data, rtol=1e-13, atol=0) and (ad @ bd).inds == (a @ b).inds

# 3. structured 1D contraction of an MPS to a dense vector (tn1d/core.py:502)
mps = qtn.MPS_rand_state(8, 5, seed=7, dtype="float64")
dense = mps.contract()
dd = on_device(mps).contract()
assert dd.inds == dense.inds and np.allclose(host(dd.data), dense.data, rtol=1e-12, atol=1e-15)

# 4. compressed boundary contraction, quimb's own driver, splits through the registered drivers
frng = np.random.default_rng(5)
tnb = qtn.TN2D_from_fill_fn(lambda shape: frng.uniform(-0.1, 1.0, size=shape), 6, 6, 2)
for chi in (2, 4):
    zb = tnb.contract_boundary(max_bond=chi)
    n0 = launched()
    zd = on_device(tnb).contract_boundary(max_bond=chi)
    assert launched() > n0 and abs(float(zd) / zb - 1) < 1e-11, (chi, float(zd), zb)
ising = qtn.TN2D_classical_ising_partition_function(8, 8, 0.44)
zi = on_device(ising).contract_boundary(max_bond=4)
gold = np.load(os.path.join(HERE, "boundary.npz"))
assert abs(float(zi) / float(gold["ising8x8_vals"][1]) - 1) < 1e-11

# 5. Tensor.split on device data: truncated SVD (several cutoff modes), QR, LQ
rng = np.random.default_rng(1)
x = qtn.Tensor(rng.normal(size=(4, 5, 6, 3)) * np.exp(-np.arange(3)), inds=["a", "b", "c", "d"])
xd = qtn.Tensor(qa.asarray(x.data), inds=x.inds)
for kw in (dict(cutoff=1e-2), dict(cutoff=1e-2, cutoff_mode="rsum2"), dict(max_bond=3, cutoff=0.0), dict(method="qr"),
           dict(method="lq"), dict(absorb="left", cutoff=1e-3), dict(absorb=None, cutoff=1e-3)):
    ref = x.split(["a", "b"], bond_ind="k", **kw)
    out = xd.split(["a", "b"], bond_ind="k", **kw)
    assert [t.shape for t in out] == [t.shape for t in ref], (kw, [t.shape for t in out], [t.shape for t in ref])
    back = out.contract(all, output_inds=x.inds, optimize="greedy")           # "k" is a hyper index when s is kept
    assert np.allclose(host(back.data), ref.contract(all, output_inds=x.inds, optimize="greedy").data,
                       rtol=1e-10, atol=1e-12), kw
    assert type(out.tensors[0].data) is qa.Array

# 6. Circuit with ``to_backend`` (circuit/core.py:88-122): an amplitude contracted on the backend
gates = [("H", 0), ("CNOT", 0, 1), ("RZ", 0.3, 1), ("CZ", 1, 2), ("U3", 0.1, 0.2, 0.3, 3), ("ISWAP", 3, 4)]
ref = qtn.Circuit(5)
ref.apply_gates(gates)
circ = qtn.Circuit(5, to_backend=qa.asarray, dtype="complex128")
circ.apply_gates(gates)
for bits in ("11000", "00000", "01100"):
    n0 = launched()
    amp = circ.amplitude(bits, simplify_sequence="")
    assert launched() > n0 and abs(complex(amp) - ref.amplitude(bits, simplify_sequence="")) < 1e-12

# 7. mode (b) of INTEGRATION.md: the tree quimb hands out (``tn.contraction_tree`` / ``contract(get="tree")``,
#    tensor_core.py:194-197) adopted by the whole-tree executor, data taken from the network in tensor order
tree = tn.contraction_tree(optimize="greedy")
ex = qa.TreeExecutor(qa.ContractionTree.from_any(tree), "float64")
m, e = ex([t.data for t in tn], strip_exponent=True)
assert abs(float(m.item()) * 10**e / want - 1) < 1e-12
assert ex.tree.contraction_cost() <= tree.contraction_cost() and ex.tree.contraction_width() == tree.contraction_width()
info = tn.contract(all, optimize="greedy", get="tree")
assert qa.ContractionTree.from_any(info).get_path() == tree.get_path()

# 8. mode (c) of INTEGRATION.md: numpy data left in place, the contraction routed by ``contract_backend``
#    (quimb/tensor/contraction.py:173-200 -> the ``backend=`` of cotengra's array_contract)
n0 = launched()
with qtn.contract_backend("quimb_amd"):
    got_c = tn.contract(all, optimize="greedy")
assert launched() - n0 == 15 and abs(float(got_c) / want - 1) < 1e-12
assert all(type(t.data) is np.ndarray for t in tn)                      # the network itself was not touched

# 9. ``tn.to(backend=...)`` via autoray.to (tensor_core.py:5312-5356)
tn_dev = tn.to(backend="quimb_amd")
assert all(type(t.data) is qa.Array for t in tn_dev) and all(type(t.data) is np.ndarray for t in tn)
assert abs(float(tn_dev.contract(all, optimize="greedy")) / want - 1) < 1e-12
tn32 = tn.to("quimb_amd-float32")
assert all(type(t.data) is qa.Array and str(t.dtype) == "float32" for t in tn32)

# 10. quimb's own ``Tensor.gate`` (tensor_core.py:3076-3166) on device data: do("tensordot") + do("transpose")
tg, tgd = tn.tensors[5], on_device(tn).tensors[5]
Gm = np.random.default_rng(2).normal(size=(tg.shape[1],) * 2)
n0 = launched()
gd = tgd.gate(qa.asarray(Gm), tg.inds[1])
assert launched() > n0 and type(gd.data) is qa.Array and gd.inds == tg.inds
assert np.allclose(host(gd.data), tg.gate(Gm, tg.inds[1]).data, rtol=1e-13, atol=0)
assert np.allclose(host(tgd.gate(qa.asarray(Gm), tg.inds[1], transpose=True, preserve_inds=False).data),
                   tg.gate(Gm, tg.inds[1], transpose=True, preserve_inds=False).data, rtol=1e-13, atol=0)

# 11. the injection point B2: a (tensordot, einsum) pair handed to cotengra through quimb's untouched kwargs
#     (tensor_core.py:293-294, :327; contraction.py:279, :291), numpy data in, every step on the backend
n0 = launched()
got_i = tn.contract(all, optimize="greedy", implementation=qa.implementation_pair())
assert launched() - n0 == 15 and abs(float(got_i) / want - 1) < 1e-12
ti = qtn.tensor_contract(tn.tensors[0], tn.tensors[1], implementation=qa.implementation_pair())
assert type(ti.data) is qa.Array and np.allclose(host(ti.data), (tn.tensors[0] @ tn.tensors[1]).data, rtol=1e-13, atol=0)

# 12. ``Circuit.amplitude`` on its DEFAULT path: ``full_simplify_("ADCRS")`` leaves a hyper-index network
#     (circuit/exact.py:421, :481-497; hyper-edge note tests/test_tensor/test_circuit/test_exact.py:136), which is
#     converted (``_maybe_convert`` -> to_backend) and contracted on the backend
import itertools  # noqa: E402
import random  # noqa: E402

random.seed(11)
nq = depth = 12                      # the circuit family of the reference's own simplification test (test_exact.py:93-139)
layers = itertools.cycle([
    lambda: [(random.choice(["X_1_2", "Y_1_2", "W_1_2"]), i) for i in range(nq)],
    lambda: [("cz", i, i + 1) for i in range(0, nq, 2)],
    lambda: [(random.choice(["X_1_2", "Y_1_2", "W_1_2"]), i) for i in range(nq)],
    lambda: [("cz", i, i + 1) for i in range(1, nq - 1, 2)],
])
g2 = [g for _, layer in zip(range(depth), layers) for g in layer()]
circ2 = qtn.Circuit(nq, to_backend=qa.asarray, dtype="complex128")
ref2 = qtn.Circuit(nq)
circ2.apply_gates(g2)
ref2.apply_gates(g2)
n_hyper = 0
for bits in ("01" * 6, "0" * 12, "110100101101"):
    tn_b = circ2.amplitude(bits, rehearse="tn")                    # simplified + converted network, not yet contracted
    assert all(type(t.data) is qa.Array for t in tn_b)
    n_hyper += sum(1 for tids in tn_b.ind_map.values() if len(tids) > 2)
    n0 = launched()
    amp = circ2.amplitude(bits, optimize="greedy")                 # simplify_sequence defaults to "ADCRS"
    want_amp = ref2.amplitude(bits, optimize="greedy")
    assert abs(complex(amp) - complex(want_amp)) < 1e-12, (bits, amp, want_amp)
    assert launched() > n0 or tn_b.num_tensors == 1
print("hyper indices met in the simplified amplitude networks:", n_hyper)
assert n_hyper > 0, "the default ADCRS path must have produced hyper-index networks for this circuit family"

print("backend launches:", dev.calls)
print("DROPIN OK")
