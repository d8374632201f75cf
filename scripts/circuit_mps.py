#!/usr/bin/env python
"""BASELINE config #2, reading A: a 53-qubit depth-10 brickwork circuit kept as an MPS with
the bond dimension capped at 32, complex64 -- ``CircuitMPS(53, max_bond=32)`` + ``gates_1D_brickwork(53, 10, "cz")``
+ ``.amplitude(b)`` in the reference (quimb/tensor/circuit/mps.py, circuit_gen.py:200).

    python scripts/circuit_mps.py [N] [depth] [max_bond]

Prints the wall time of applying the gates on the device and compares amplitudes / norm with the same circuit
run in complex128 on the numpy plan interpreter of the tests (same truncation rule, higher precision)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import quimb_amd as qa
import quimb_amd.device as qd
from quimb_amd.circuit import CircuitMPS

N = int(sys.argv[1]) if len(sys.argv) > 1 else 53
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 10
chi = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rng = np.random.default_rng(7)
gates = []
pairs = []
for d in range(depth):                       # gates_1D_brickwork: even layer, odd layer, U3s injected around them
    pairs += [(i, i + 1) for i in range(0, N - 1, 2)] + [(i, i + 1) for i in range(1, N - 1, 2)]
for q in range(N):
    gates.append(("U3", *rng.uniform(0, 2 * np.pi, 3), q))
for (a, b) in pairs:
    gates.append(("CZ", a, b))
    gates.append(("U3", *rng.uniform(0, 2 * np.pi, 3), a))
    gates.append(("U3", *rng.uniform(0, 2 * np.pi, 3), b))
bits = ["01" * (N // 2) + "0" * (N % 2), "0" * N, "1" * N]

dev = qa.default_device()
best = None
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cm = CircuitMPS(N, max_bond=chi, dtype="complex64").apply_gates(gates)
    amps = [cm.amplitude(b) for b in bits]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print(f"CircuitMPS N={N} depth={depth} max_bond={chi} complex64: {len(gates)} gates, {best:.2f} s "
      f"({best / len(gates) * 1e3:.2f} ms per gate), max bond {cm.max_bond_dim()}, norm {cm.norm():.6f}, "
      f"fidelity estimate {cm.fidelity_estimate():.4f}")
from emu_device import EmuDevice
old = qd._DEFAULT
qd.set_default_device(EmuDevice())
t0 = time.perf_counter()
ref = CircuitMPS(N, max_bond=chi, dtype="complex128").apply_gates(gates)
ramps = [ref.amplitude(b) for b in bits]
dt = time.perf_counter() - t0
qd.set_default_device(old)
print(f"numpy plan interpreter, complex128: {dt:.2f} s, norm {ref.norm():.6f}")
for b, a, r in zip(bits, amps, ramps):
    print(f"  <{b[:8]}...|psi> = {a:.6e}   ref {r:.6e}   |diff|/|ref| = {abs(a - r) / abs(r):.2e}")
