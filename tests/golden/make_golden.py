#!/usr/bin/env python
"""Generate golden vectors by running the REAL quimb sources from /root/reference.

quimb's hard imports that are absent from this image (autoray, cotengra, numba,
cytoolz/toolz -- SURVEY.md section 0.3) are satisfied by the minimal numpy-only
stand-ins in ``tests/golden/_shims`` (dispatch + a restated pairwise contraction
loop); everything else -- Tensor / TensorNetwork bookkeeping, tensor_contract's
output-index / scalar / tag / exponent rules, the builders (TN2D_rand,
MPS_rand_state, TN2D_classical_ising_partition_function, PEPS.rand), Tensor.fuse /
transpose / isel, contract_structured for 1D -- is quimb's own code.

    PYTHONPATH=tests/golden/_shims:/root/reference python tests/golden/make_golden.py

writes tests/golden/*.npz (committed).  /root/reference only exists in the build
container, so nothing at test time imports it.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, "/root/reference")

import quimb as qu  # noqa: E402
import quimb.tensor as qtn  # noqa: E402


def enc_inds(inds):
    return json.dumps([list(map(str, t)) for t in inds])


def save_network(name, tn, extra):
    arrays = [np.asarray(t.data) for t in tn]
    inds = [t.inds for t in tn]
    tags = [list(t.tags) for t in tn]
    out = {f"a{i}": a for i, a in enumerate(arrays)}
    out["inds"] = enc_inds(inds)
    out["tags"] = json.dumps(tags)
    out["n"] = len(arrays)
    for k, v in extra.items():
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: (np.shape(v) if hasattr(v, "shape") else v) for k, v in extra.items() if k != "inds"})


def main():
    rng = np.random.default_rng(1234)

    # 1. pairwise tensor_contract cases (reference tests/test_tensor/test_tensor_core.py:434-512)
    a = qtn.Tensor(rng.normal(size=(2, 3, 4)), inds=["i0", "i1", "i2"], tags="red")
    b = qtn.Tensor(rng.normal(size=(3, 4, 5)), inds=["i1", "i2", "i3"], tags="blue")
    c = qtn.Tensor(rng.normal(size=(5, 2, 6)), inds=["i3", "i0", "i4"], tags="blue")
    ab = a @ b
    abc = qtn.tensor_contract(a, b, c)
    b2 = qtn.Tensor(rng.normal(size=(3, 4, 2)), inds=["i1", "i2", "i0"])
    scalar = a @ b2
    outer = a @ qtn.Tensor(b.data, inds=["j5", "j4", "j3"])
    np.savez_compressed(
        os.path.join(HERE, "pairwise.npz"),
        a=a.data, b=b.data, c=c.data, b2=b2.data,
        ab=ab.data, ab_inds=json.dumps(list(ab.inds)),
        abc=abc.data, abc_inds=json.dumps(list(abc.inds)), abc_tags=json.dumps(list(abc.tags)),
        scalar=np.asarray(scalar), scalar_is_float=isinstance(scalar, float),
        outer=outer.data, outer_inds=json.dumps(list(outer.inds)),
    )
    print("wrote pairwise")

    # 2. random 2D lattice (TN2D_rand, index order l,r,u,d), exact contraction
    tn = qtn.TN2D_rand(4, 4, 3, seed=42, dtype="float64")
    z = tn.contract(all, optimize="greedy")
    m, e = tn.contract(all, optimize="greedy", strip_exponent=True)
    save_network("tn2d_rand_4x4_D3", tn, dict(value=np.asarray(z), mantissa=np.asarray(m), exponent=np.asarray(e)))

    # 3. classical Ising partition function 6x6 (builder tensor_builder.py:2687)
    tn = qtn.TN2D_classical_ising_partition_function(6, 6, 0.44)
    z = tn.contract(all, optimize="greedy")
    save_network("ising_6x6_b044", tn, dict(value=np.asarray(z)))

    # 4. MPS -> dense vector via quimb's structured contraction (tn1d/core.py:502)
    mps = qtn.MPS_rand_state(8, 5, seed=7, dtype="float64")
    dense = mps.contract()
    save_network("mps_L8_chi5", mps, dict(dense=np.asarray(dense.data), dense_inds=json.dumps(list(dense.inds))))

    # 5. PEPS amplitude: isel every physical index, contract the single-layer network
    peps = qtn.PEPS.rand(3, 3, 4, seed=11, dtype="float64")
    bits = [0, 1, 1, 0, 1, 0, 0, 1, 1]
    sel = {peps.site_ind(i, j): bits[i * 3 + j] for i in range(3) for j in range(3)}
    amp_tn = peps.isel(sel)
    amp = amp_tn.contract(all, optimize="greedy")
    save_network("peps_3x3_D4_amp", amp_tn, dict(value=np.asarray(amp)))

    # 6. hyper-index network with explicit output_inds
    hin = [("a", "x"), ("b", "x"), ("c", "x", "y"), ("y", "d"), ("a", "b"), ("c", "d", "y")]
    size = dict(a=3, b=4, c=2, d=5, x=3, y=4)
    ts = [qtn.Tensor(rng.normal(size=[size[i] for i in t]), inds=t) for t in hin]
    htn = qtn.TensorNetwork(ts)
    h0 = htn.contract(all, output_inds=(), optimize="greedy")
    hx = htn.contract(all, output_inds=("y", "x"), optimize="greedy")
    save_network("hyper_net", htn, dict(value=np.asarray(h0), yx=np.asarray(hx.data)))

    # 7. fuse / transpose / isel of a Tensor (array_ops.py:95-182, tensor_core.py:3252-3298)
    t = qtn.Tensor(rng.normal(size=(2, 3, 4, 5, 6)), inds=["a", "b", "c", "d", "e"])
    f1 = t.fuse({"ce": ["e", "c"], "da": ["d", "a"]})
    f2 = t.fuse({"bc": ["b", "c"]})
    tr = t.transpose("e", "c", "a", "d", "b")
    sl = t.isel({"c": 2, "a": 1})
    np.savez_compressed(
        os.path.join(HERE, "layout.npz"),
        t=t.data, f1=f1.data, f1_inds=json.dumps(list(f1.inds)), f2=f2.data, f2_inds=json.dumps(list(f2.inds)),
        tr=tr.data, sl=sl.data, sl_inds=json.dumps(list(sl.inds)),
    )
    print("wrote layout")

    # 8. cut_iter sliced-sum identity (tensor_core.py:9291-9328)
    tn = qtn.TN2D_rand(3, 3, 3, seed=5, dtype="float64")
    full = tn.contract(all, optimize="greedy")
    cut = [ix for ix in tn.inner_inds()][:2]
    parts = [stn.contract(all, optimize="greedy") for stn in tn.cut_iter(*cut)]
    save_network("tn2d_cut_3x3_D3", tn, dict(value=np.asarray(full), cut=json.dumps(list(cut)), parts=np.asarray(parts)))
    # 9. boundary contraction (TensorNetwork2D.contract_boundary, tn2d/core.py:2502; the accuracy test
    #    tests/test_tensor/test_tn2d/test_core.py:241-274 uses the same uniform(-0.1, 1) fill)
    def site_arrays(tn, Lx, Ly):
        """Row-major site data re-ordered to l, r, u (row i+1), d (row i-1) by bond NAME (no layout assumed)."""
        out = []
        for i in range(Lx):
            for j in range(Ly):
                t = tn[tn.site_tag(i, j)]
                order = []
                for (ii, jj) in ((i, j - 1), (i, j + 1), (i + 1, j), (i - 1, j)):
                    if 0 <= ii < Lx and 0 <= jj < Ly:
                        (b,) = qtn.bonds(t, tn[tn.site_tag(ii, jj)])
                        order.append(b)
                out.append(np.asarray(t.transpose(*order).data))
        return out

    cases = {}
    for name, Lx, Ly, D, chis in (("u8x8D2", 8, 8, 2, (2, 4)), ("u6x6D3", 6, 6, 3, (3, 6)),
                                   ("u4x7D3", 4, 7, 3, (4,)), ("u7x4D3", 7, 4, 3, (4,))):
        frng = np.random.default_rng(hash(name) % 1000 if False else len(name) * 100 + Lx * 10 + Ly)
        tn = qtn.TN2D_from_fill_fn(lambda shape: frng.uniform(-0.1, 1.0, size=shape), Lx, Ly, D)
        cases[name] = (tn, Lx, Ly, chis)
    cases["ising8x8"] = (qtn.TN2D_classical_ising_partition_function(8, 8, 0.44), 8, 8, (2, 4))
    out = {"names": json.dumps(sorted(cases))}
    for name, (tn, Lx, Ly, chis) in cases.items():
        arrs = site_arrays(tn, Lx, Ly)
        for k, a in enumerate(arrs):
            out[f"{name}_a{k}"] = a
        out[f"{name}_shape"] = np.array([Lx, Ly])
        out[f"{name}_exact"] = np.asarray(tn.contract(all, optimize="greedy"))
        out[f"{name}_chis"] = np.array(chis)
        out[f"{name}_vals"] = np.array([tn.contract_boundary(max_bond=chi) for chi in chis])
        m, e = tn.contract_boundary(max_bond=chis[-1], strip_exponent=True)
        out[f"{name}_stripped"] = np.array([m, e])
        print(name, out[f"{name}_exact"], out[f"{name}_vals"], out[f"{name}_stripped"])
    np.savez_compressed(os.path.join(HERE, "boundary.npz"), **out)
    print("wrote boundary")
    # 10. DMRG2 (tn1d/dmrg.py): the reference's MPO tensors, its per-sweep energies for a fixed schedule and
    #     the dense ground energy (tests/test_tensor/test_tn1d/test_dmrg.py:240-311)
    out = {}
    for name, L, kw in (("heis10", 10, {}), ("heis6_bz", 6, {"bz": 0.3})):
        H = qtn.MPO_ham_heis(L, **kw)
        for k in range(L):
            out[f"{name}_w{k}"] = np.asarray(H[k].data)      # first (r,k,b) / bulk (l,r,k,b) / last (l,k,b)
        out[f"{name}_dense"] = np.asarray(H.to_dense())
        dm = qtn.DMRG2(H, bond_dims=[8, 16, 32], cutoffs=1e-10)
        out[f"{name}_converged"] = dm.solve(tol=1e-9, max_sweeps=8)
        out[f"{name}_energies"] = np.array(dm.energies)
        out[f"{name}_e0"] = np.linalg.eigvalsh(out[f"{name}_dense"])[0]
        out[f"{name}_max_bond"] = dm.state.max_bond()
        print(name, out[f"{name}_energies"], out[f"{name}_e0"], out[f"{name}_max_bond"])
    builder = qtn.SpinHam1D(1 / 2)
    builder[0, 1] += 1.0, "Z", "Z"
    H = builder.build_mpo(2)
    out["zz2_w0"], out["zz2_w1"] = np.asarray(H[0].data), np.asarray(H[1].data)
    dm = qtn.DMRG2(H)
    dm.solve()
    out["zz2_energy"] = dm.energy
    np.savez_compressed(os.path.join(HERE, "dmrg.npz"), **out)
    print("wrote dmrg", out["zz2_energy"])
    # 11. tensor_split / array_split policy (tensor_core.py:390, decomp.py:35): kept rank, singular values and
    #     the (gauge-independent) product of the factors for a grid of cutoff modes / absorbs / methods
    srng = np.random.default_rng(77)
    spec = np.exp(-0.9 * np.arange(12))
    ul, _ = np.linalg.qr(srng.normal(size=(24, 12)))
    vr, _ = np.linalg.qr(srng.normal(size=(30, 12)))
    x = (ul * spec) @ vr.T
    t = qtn.Tensor(x.reshape(4, 6, 5, 6), inds=["a", "b", "c", "d"])
    out = {"x": x}
    cases = []
    for ci, kw in enumerate([
        dict(cutoff=1e-10), dict(cutoff=1e-3), dict(cutoff=1e-3, cutoff_mode="abs"),
        dict(cutoff=1e-3, cutoff_mode="sum2"), dict(cutoff=1e-3, cutoff_mode="rsum2"),
        dict(cutoff=1e-2, cutoff_mode="sum1"), dict(cutoff=1e-2, cutoff_mode="rsum1"),
        dict(cutoff=0.0, max_bond=5), dict(cutoff=1e-3, max_bond=3), dict(cutoff=1e-3, cutoff_mode="rsum2", renorm=True),
        dict(cutoff=0.0, max_bond=4, renorm=2), dict(cutoff=1e-4, method="eig"), dict(cutoff=0.0, method="qr", absorb="right"),
        dict(cutoff=0.0, method="lq", absorb="left"), dict(cutoff=1e-3, absorb="left"), dict(cutoff=1e-3, absorb="right"),
        dict(cutoff=1e-3, absorb=None),
    ]):
        import warnings as _w
        with _w.catch_warnings():
            _w.simplefilter("ignore")
            parts = t.split(["a", "b"], get="arrays", **kw)
        if len(parts) == 3:
            l, sv, r = parts
            prod_ = np.einsum("abk,k,kcd->abcd", l, sv, r)
            out[f"s{ci}"] = sv
        else:
            l, r = parts
            prod_ = np.einsum("abk,kcd->abcd", l, r)
        out[f"p{ci}"] = prod_
        out[f"k{ci}"] = l.shape[-1]
        cases.append(kw)
        print("split", kw, "rank", out[f"k{ci}"])
    out["values"] = np.asarray(t.split(["a", "b"], get="values", method="svd"))   # the whole spectrum, untruncated
    out["cases"] = json.dumps(cases)
    np.savez_compressed(os.path.join(HERE, "split.npz"), **out)
    print("wrote split")
    # 12. circuits (circuit/exact.py:417 Circuit.amplitude / to_dense; circuit/mps.py CircuitMPS): gate lists as
    #     plain data, dense states and amplitudes of the real quimb
    def plain(gates):
        rows = []
        for g in gates:
            if hasattr(g, "label"):
                rows.append([g.label, [float(p) for p in g.params], [int(q) for q in g.qubits]])
            else:
                name, *rest = g
                npar = len(rest) - (2 if name.upper() in ("CZ", "CNOT", "CX", "CY", "ISWAP", "SWAP", "FSIM", "RZZ") else 1)
                rows.append([name.upper(), [float(p) for p in rest[:npar]], [int(q) for q in rest[npar:]]])
        return rows

    crng = np.random.default_rng(3)
    N = 5
    gates = []
    for d in range(3):
        for q in range(N):
            gates.append(("U3", *crng.uniform(0, 2 * np.pi, 3), q))
        for q in range(d % 2, N - 1, 2):
            gates.append(("CZ", q, q + 1))
    gates += [("H", 0), ("CNOT", 0, 3), ("RZ", 0.3, 2), ("ISWAP", 1, 4), ("T", 2), ("S", 1), ("RX", 0.7, 4),
              ("RY", 1.1, 0), ("SWAP", 2, 3), ("X", 1), ("Y", 2), ("Z", 3), ("FSIM", 0.4, 0.9, 0, 1), ("CY", 4, 2),
              ("RZZ", 0.45, 1, 3)]
    circ = qtn.Circuit(N)
    circ.apply_gates(gates)
    out = {"exact_gates": json.dumps(plain(gates)), "exact_N": N,
           "exact_dense": np.asarray(circ.to_dense()).reshape(-1)}
    bits = ["01101", "00000", "11111", "10010"]
    out["exact_bits"] = json.dumps(bits)
    out["exact_amps"] = np.array([circ.amplitude(b, simplify_sequence="") for b in bits])
    from quimb.tensor import circuit_gen
    for name, n, depth, chi in (("mps_exact", 8, 4, None), ("mps_chi4", 8, 6, 4), ("mps_chi8_nonlocal", 7, 3, 8)):
        gl = list(circuit_gen.gates_1D_brickwork(n, depth, seed=11))
        if name.endswith("nonlocal"):
            gl += [("CZ", 0, 4), ("CNOT", 5, 1), ("ISWAP", 6, 2), ("H", 3)]
        cm = qtn.CircuitMPS(n, max_bond=chi)
        cm.apply_gates(gl)
        out[f"{name}_gates"] = json.dumps(plain(cm.gates))
        out[f"{name}_N"] = n
        out[f"{name}_chi"] = -1 if chi is None else chi
        out[f"{name}_dense"] = np.asarray(cm.psi.to_dense()).reshape(-1)
        mbits = ["01" * (n // 2) + "0" * (n % 2), "0" * n, "1" * n]
        out[f"{name}_bits"] = json.dumps(mbits)
        out[f"{name}_amps"] = np.array([cm.amplitude(b) for b in mbits])
        out[f"{name}_max_bond"] = cm.psi.max_bond()
        print(name, out[f"{name}_amps"], out[f"{name}_max_bond"], np.linalg.norm(out[f"{name}_dense"]))
    np.savez_compressed(os.path.join(HERE, "circuit.npz"), **out)
    print("wrote circuit", out["exact_amps"])
    print("quimb version:", qu.__version__)


if __name__ == "__main__":
    main()
