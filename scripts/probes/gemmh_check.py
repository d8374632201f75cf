"""The split-product joins (quimb_amd.Options.join_arith = "f16x3", csrc/gemmh.hip) through the Python boundary: GEMM-shaped
pairs of ragged sizes and several index orders against fp64 numpy, beside the same pair on the default fp32 MFMA path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quimb_amd as qa

rng = np.random.default_rng(5)
dev = qa.default_device()
bad = 0
cases = [
    # (a inds, a shape, b inds, b shape, out inds)
    ("km", (512, 700), "kn", (512, 300), "mn"),
    ("km", (777, 1300), "kn", (777, 516), "nm"),
    ("kab", (1296, 36, 36), "kcd", (1296, 6, 216), "abcd"),
    ("kab", (1000, 20, 30), "kcd", (1000, 40, 10), "cadb"),     # not both bundles contiguous in C: whatever the planner picks
    ("km", (7776, 1944), "kn", (7776, 972), "mn"),
]
for fill in ("uniform", "signed", "wide"):
    for ai, ash, bi, bsh, oi in cases:
        if fill == "uniform":
            a, b = rng.uniform(-0.1, 1, ash), rng.uniform(-0.1, 1, bsh)
        elif fill == "signed":
            a, b = rng.normal(size=ash), rng.normal(size=bsh) * 1e-4
        else:
            a, b = rng.lognormal(0, 3, ash) * rng.choice([-1, 1], ash), rng.lognormal(0, 3, bsh)
        a, b = a.astype(np.float32), b.astype(np.float32)
        want = np.einsum(f"{ai},{bi}->{oi}", a.astype(np.float64), b.astype(np.float64), optimize=True)
        scale = np.abs(want).max()
        res = {}
        for mode in ("f32", "f16x3-all"):
            with qa.exec_options(join_arith=mode):
                got = qa.einsum(f"{ai},{bi}->{oi}", qa.asarray(a), qa.asarray(b))
                res[mode] = np.abs(qa.to_numpy(got).astype(np.float64) - want).max() / scale
        with qa.exec_options(join_arith="f16x3-all"):
            from quimb_amd.pairwise import plan_pair
            step = plan_pair(tuple(ai), ash, tuple(bi), bsh, tuple(oi), True)
            name = dev.describe_pair(dev.compile_pair(step.spec, np.dtype("float32")))
        res["f16x3"] = res["f16x3-all"]
        ok = res["f16x3"] < max(2e-6, 2 * res["f32"])
        bad += not ok
        print(f"{fill:8s} {ai}{ash} x {bi}{bsh} -> {oi}: {name:34s} max-norm err vs fp64: f16x3 {res['f16x3']:.2e}, fp32 MFMA path {res['f32']:.2e}  {'ok' if ok else '** FAILED **'}")
# complex operands: 2 x 2 real blocks on the real kernels (ops.complex_expand): re / im interleaved along K -- centred per parity of k
for (K, M, N) in ((512, 384, 640), (1296, 1296, 216)):
    a = (rng.uniform(-0.1, 1, (K, M)) + 1j * rng.uniform(-0.1, 1, (K, M))).astype(np.complex64)
    b = (rng.uniform(-0.1, 1, (K, N)) + 1j * rng.uniform(-1, 0.1, (K, N))).astype(np.complex64)
    want = a.astype(np.complex128).T @ b.astype(np.complex128)
    res = {}
    for mode in ("f32", "f16x3-all"):
        with qa.exec_options(join_arith=mode):
            got = qa.to_numpy(qa.einsum("km,kn->mn", qa.asarray(a), qa.asarray(b)))
        res["f16x3" if mode != "f32" else mode] = np.abs(got.astype(np.complex128) - want).max() / np.abs(want).max()
    ok = res["f16x3"] < max(2e-6, 2 * res["f32"])
    bad += not ok
    print(f"complex64 km({K}, {M}) x kn({K}, {N}) -> mn: max-norm err vs complex128: f16x3 {res['f16x3']:.2e}, fp32 MFMA path {res['f32']:.2e}  {'ok' if ok else '** FAILED **'}")
# timing of whole calls (split passes included) for the four operand layouts of a square product, both arithmetics
import torch
for n in (4096, 8192):
    for ai, bi in (("km", "kn"), ("mk", "kn"), ("km", "nk"), ("mk", "nk")):
        a = qa.asarray(rng.uniform(-0.1, 1, (n, n)).astype(np.float32))
        b = qa.asarray(rng.uniform(-0.1, 1, (n, n)).astype(np.float32))
        line = f"{ai},{bi}->mn {n}^3:"
        for mode in ("f32", "f16x3-all"):
            with qa.exec_options(join_arith=mode):
                for _ in range(2):
                    qa.einsum(f"{ai},{bi}->mn", a, b)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    qa.einsum(f"{ai},{bi}->mn", a, b)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 5
                from quimb_amd.pairwise import plan_pair
                step = plan_pair(tuple(ai), (n, n), tuple(bi), (n, n), ("m", "n"), True)
                name = dev.describe_pair(dev.compile_pair(step.spec, np.dtype("float32")))
            line += f"  {mode}: {ms:.3f} ms = {2 * n**3 / ms * 1e-9:.0f} TFLOP/s ({name.split('<')[0]})"
        print(line)
        del a, b
print("FAILED" if bad else "all ok")
sys.exit(1 if bad else 0)
