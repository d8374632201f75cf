"""Quantum circuits as callers of the contraction path.

``Circuit``     -- the exact simulator's data model: every gate is one small tensor of a network
                   (quimb/tensor/circuit/core.py ``CircuitBase.apply_gate``, circuit/exact.py:417-501 ``amplitude``,
                   ``to_dense``); an amplitude is ONE contraction, which ``array_contract`` routes to the
                   one-launch ``MicroTree`` walker (many bitstrings: one launch for all of them).
``CircuitMPS``  -- the state kept as a matrix product state (circuit/mps.py): one-qubit gates are absorbed into a
                   site, two-qubit gates on neighbours are contracted in and split again with ``max_bond`` /
                   ``cutoff`` (``tensor_split``'s relative cutoff), distant qubits are brought together with SWAPs
                   ("swap+split").  The orthogonality centre follows the gates, so every truncation is the optimal one.

Gate matrices follow the reference's conventions (circuit/gates.py): qubit 0 is the most significant bit of a
dense state, two-qubit matrices act on |q_a q_b> in that order.
"""

import cmath
import math

import numpy as np

from . import linalg, ops
from .array import asarray
from .contract import array_contract, array_contract_expression
from .split import svals_to_keep

_SQ2 = 1.0 / math.sqrt(2.0)


def _u3(theta, phi, lamda):
    c, s = math.cos(theta / 2), math.sin(theta / 2)
    return np.array([[c, -cmath.exp(1j * lamda) * s],
                     [cmath.exp(1j * phi) * s, cmath.exp(1j * (phi + lamda)) * c]])


def _rot(axis, theta):
    c, s = math.cos(theta / 2), math.sin(theta / 2)
    if axis == "x":
        return np.array([[c, -1j * s], [-1j * s, c]])
    if axis == "y":
        return np.array([[c, -s], [s, c]], dtype=complex)
    return np.array([[cmath.exp(-0.5j * theta), 0], [0, cmath.exp(0.5j * theta)]])


def _controlled(u):
    g = np.eye(4, dtype=complex)
    g[2:, 2:] = u
    return g


_X = np.array([[0, 1], [1, 0]], dtype=complex)
_Y = np.array([[0, -1j], [1j, 0]])
_Z = np.array([[1, 0], [0, -1]], dtype=complex)

#: name -> (number of parameters, number of qubits, matrix builder)
GATES = {
    "H": (0, 1, lambda: np.array([[_SQ2, _SQ2], [_SQ2, -_SQ2]], dtype=complex)),
    "X": (0, 1, lambda: _X), "Y": (0, 1, lambda: _Y), "Z": (0, 1, lambda: _Z),
    "S": (0, 1, lambda: np.diag([1, 1j])), "T": (0, 1, lambda: np.diag([1, cmath.exp(0.25j * math.pi)])),
    "IDEN": (0, 1, lambda: np.eye(2, dtype=complex)),
    "RX": (1, 1, lambda t: _rot("x", t)), "RY": (1, 1, lambda t: _rot("y", t)), "RZ": (1, 1, lambda t: _rot("z", t)),
    "U3": (3, 1, _u3),
    "U2": (2, 1, lambda phi, lam: _u3(math.pi / 2, phi, lam)),
    "U1": (1, 1, lambda lam: np.diag([1, cmath.exp(1j * lam)])),
    "CX": (0, 2, lambda: _controlled(_X)), "CNOT": (0, 2, lambda: _controlled(_X)),
    "CY": (0, 2, lambda: _controlled(_Y)), "CZ": (0, 2, lambda: _controlled(_Z)),
    "SWAP": (0, 2, lambda: np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=complex)),
    "ISWAP": (0, 2, lambda: np.array([[1, 0, 0, 0], [0, 0, 1j, 0], [0, 1j, 0, 0], [0, 0, 0, 1]])),
    "FSIM": (2, 2, lambda theta, phi: np.array(
        [[1, 0, 0, 0], [0, math.cos(theta), -1j * math.sin(theta), 0],
         [0, -1j * math.sin(theta), math.cos(theta), 0], [0, 0, 0, cmath.exp(-1j * phi)]])),
    "RZZ": (1, 2, lambda g: np.diag([cmath.exp(-0.5j * g), cmath.exp(0.5j * g), cmath.exp(0.5j * g),
                                     cmath.exp(-0.5j * g)])),
}


def parse_gate(gate):
    """``("U3", theta, phi, lam, q)`` / ``("CZ", a, b)`` (the reference's tuple form, circuit/gates.py
    ``parse_to_gate``) or ``[name, params, qubits]`` -> (name, params, qubits, matrix)."""
    name = str(gate[0]).upper()
    if name not in GATES:
        raise ValueError(f"unknown gate {gate[0]!r}")
    npar, nq, build = GATES[name]
    if len(gate) == 3 and isinstance(gate[1], (list, tuple)) and isinstance(gate[2], (list, tuple)):
        params, qubits = tuple(gate[1]), tuple(gate[2])
    else:
        rest = tuple(gate[1:])
        params, qubits = rest[:npar], rest[npar:]
    if len(params) != npar or len(qubits) != nq:
        raise ValueError(f"gate {name} takes {npar} parameter(s) and {nq} qubit(s), got {gate!r}")
    qubits = tuple(int(q) for q in qubits)
    if len(set(qubits)) != nq:
        raise ValueError(f"gate {name} needs distinct qubits, got {qubits}")
    return name, tuple(float(p) for p in params), qubits, np.asarray(build(*params), dtype=np.complex128)


def _bits(b, n):
    b = [int(c) for c in b] if isinstance(b, str) else [int(c) for c in b]
    if len(b) != n or any(c not in (0, 1) for c in b):
        raise ValueError(f"need a bitstring of {n} zeros and ones")
    return b


class Circuit:
    """Exact simulation by contraction of the gate network."""

    def __init__(self, N, dtype="complex128"):
        self.N = int(N)
        if self.N < 1:
            raise ValueError("need at least one qubit")
        self.dtype = np.dtype(dtype)
        if self.dtype.kind != "c":
            raise ValueError("circuit amplitudes are complex: use complex64 or complex128")
        self.gates = []
        zero = np.array([1.0, 0.0], dtype=self.dtype)
        self._arrays = [asarray(zero) for _ in range(self.N)]
        self._inputs = [(("k", q, 0),) for q in range(self.N)]
        self._wire = [0] * self.N                      # current wire segment of every qubit
        self._exprs = {}

    def apply_gate(self, *gate):
        if len(gate) == 1 and not isinstance(gate[0], str):
            gate = gate[0]
        name, params, qubits, mat = parse_gate(gate)
        if any(not 0 <= q < self.N for q in qubits):
            raise ValueError(f"qubit out of range in {gate!r}")
        nq = len(qubits)
        ins = tuple(("k", q, self._wire[q]) for q in qubits)
        for q in qubits:
            self._wire[q] += 1
        outs = tuple(("k", q, self._wire[q]) for q in qubits)
        self._arrays.append(asarray(mat.reshape((2,) * (2 * nq)).astype(self.dtype)))
        self._inputs.append(outs + ins)
        self.gates.append((name, params, qubits))
        self._exprs.clear()
        return self

    def apply_gates(self, gates):
        for g in gates:
            self.apply_gate(g)
        return self

    def _out_inds(self):
        return tuple(("k", q, self._wire[q]) for q in range(self.N))

    def to_dense(self, optimize="greedy"):
        """The full state vector, qubit 0 most significant (``Circuit.to_dense``) -- small N only."""
        out = array_contract(self._arrays, self._inputs, self._out_inds(), optimize=optimize)
        return out.reshape((2**self.N,))

    def _amp_network(self):
        inputs = list(self._inputs) + [(ix,) for ix in self._out_inds()]
        return inputs

    def amplitude(self, b, optimize="greedy"):
        """<b|psi> as one contraction (``Circuit.amplitude``, circuit/exact.py:417-501, without the optional
        simplification passes)."""
        bits = _bits(b, self.N)
        kets = [asarray(np.eye(2, dtype=self.dtype)[c]) for c in bits]
        z = array_contract(self._arrays + kets, self._amp_network(), (), optimize=optimize)
        return complex(np.asarray(z.to_numpy() if hasattr(z, "to_numpy") else z).item())

    def amplitudes(self, bitstrings, optimize="greedy"):
        """Many amplitudes of the SAME circuit: one contraction tree, and -- when the tree qualifies for the
        ``MicroTree`` walker -- one kernel launch for all of them (the bras are the only inputs that differ)."""
        rows = [_bits(b, self.N) for b in bitstrings]
        if not rows:
            return np.zeros(0, dtype=self.dtype)
        inputs = self._amp_network()
        e0, e1 = (asarray(np.eye(2, dtype=self.dtype)[c]) for c in (0, 1))
        arrays = self._arrays + [e0] * self.N
        key = ("amp", optimize)
        expr = self._exprs.get(key)
        if expr is None:
            expr = array_contract_expression(inputs, (), shapes=[a.shape for a in arrays], optimize=optimize,
                                             dtype=self.dtype, cache=False)
            self._exprs[key] = expr
        micro = getattr(expr, "_micro", None)
        if micro is not None:
            bound = micro.bind(arrays)
            n0 = len(self._arrays)
            choice = np.asarray(rows, dtype=np.int64)
            out = bound.batch({n0 + q: ((e0, e1), choice[:, q]) for q in range(self.N)})
            return out.to_numpy().reshape(-1)
        res = []
        for r in rows:
            kets = [e1 if c else e0 for c in r]
            z = expr(*(self._arrays + kets))
            res.append(complex(np.asarray(z.to_numpy() if hasattr(z, "to_numpy") else z).item()))
        return np.asarray(res, dtype=self.dtype)


class CircuitMPS:
    """State kept as an MPS; site tensors are (left, phys, right)."""

    def __init__(self, N, max_bond=None, cutoff=1e-10, dtype="complex128"):
        self.N = int(N)
        if self.N < 2:
            raise ValueError("need at least two qubits")
        self.dtype = np.dtype(dtype)
        if self.dtype.kind != "c":
            raise ValueError("use complex64 or complex128")
        self.max_bond, self.cutoff = max_bond, cutoff
        z = np.zeros((1, 2, 1), dtype=self.dtype)
        z[0, 0, 0] = 1.0
        self._A = [asarray(z) for _ in range(self.N)]
        self._center = 0                               # a product state is canonical about any site
        self.gates = []
        self.truncation_errors = []                    # discarded fraction of the weight, per split

    # ---- gauge ---------------------------------------------------------------------------------------------
    def _shift_center(self, site):
        A = self._A
        while self._center < site:
            i = self._center
            l, p, r = A[i].shape
            q, rr = linalg.qr(A[i].reshape((l * p, r)))
            A[i] = q.reshape((l, p, q.shape[1]))
            A[i + 1] = ops.tensordot(rr, A[i + 1], axes=([1], [0]))
            self._center += 1
        while self._center > site:
            i = self._center
            l, p, r = A[i].shape
            q, rr = linalg.qr(ops.transpose(A[i].reshape((l, p * r)), (1, 0)))     # A = (R^T)(Q^T)
            A[i] = ops.transpose(q, (1, 0)).reshape((q.shape[1], p, r))
            A[i - 1] = ops.tensordot(A[i - 1], ops.transpose(rr, (1, 0)), axes=([2], [0]))
            self._center -= 1

    # ---- gates ---------------------------------------------------------------------------------------------
    def _apply_1q(self, mat, q):
        g = asarray(mat.astype(self.dtype))
        self._A[q] = array_contract([g, self._A[q]], [("P", "p"), ("l", "p", "r")], ("l", "P", "r"))

    def _apply_2q_adjacent(self, mat, i, flipped=False):
        """Gate on sites (i, i + 1); ``flipped``: the matrix is given for the qubit order (i + 1, i)."""
        self._shift_center(i)
        A = self._A
        g = mat.reshape(2, 2, 2, 2)
        if flipped:
            g = g.transpose(1, 0, 3, 2)
        g = asarray(np.ascontiguousarray(g).astype(self.dtype))
        theta = array_contract([g, A[i], A[i + 1]],
                               [("P", "Q", "p", "q"), ("l", "p", "m"), ("m", "q", "r")], ("l", "P", "Q", "r"))
        l, _, _, r = theta.shape
        u, s, vh = linalg.svd(theta.reshape((l * 2, 2 * r)))
        sh = s.to_numpy().astype(np.float64)
        k = svals_to_keep(sh, self.cutoff, "rel", self.max_bond)
        tot = float((sh**2).sum())
        self.truncation_errors.append(float((sh[k:] ** 2).sum()) / tot if tot > 0 else 0.0)
        A[i] = u[:, :k].reshape((l, 2, k))
        A[i + 1] = ops.multiply(vh[:k, :], asarray(sh[:k].astype(self.dtype))[:, None]).reshape((k, 2, r))
        self._center = i + 1

    def apply_gate(self, *gate):
        if len(gate) == 1 and not isinstance(gate[0], str):
            gate = gate[0]
        name, params, qubits, mat = parse_gate(gate)
        if any(not 0 <= q < self.N for q in qubits):
            raise ValueError(f"qubit out of range in {gate!r}")
        if len(qubits) == 1:
            self._apply_1q(mat, qubits[0])
        else:
            a, b = qubits
            lo, hi = min(a, b), max(a, b)
            swap = GATES["SWAP"][2]()
            for j in range(hi - 1, lo, -1):            # bring the far qubit next to the near one ...
                self._apply_2q_adjacent(swap, j)
            self._apply_2q_adjacent(mat, lo, flipped=a > b)
            for j in range(lo + 1, hi):                # ... and take it back
                self._apply_2q_adjacent(swap, j)
        self.gates.append((name, params, qubits))
        return self

    def apply_gates(self, gates):
        for g in gates:
            self.apply_gate(g)
        return self

    # ---- read-out ------------------------------------------------------------------------------------------
    def max_bond_dim(self):
        return max(a.shape[2] for a in self._A[:-1])

    def amplitude(self, b):
        """<b|psi>: select the physical index of every site and multiply the chain of matrices
        (``CircuitMPS.amplitude``, circuit/mps.py:192-208)."""
        bits = _bits(b, self.N)
        env = None
        for a, c in zip(self._A, bits):
            m = a[:, c, :]
            env = m if env is None else ops.tensordot(env, m, axes=([1], [0]))
        return complex(env.reshape(()).item())

    def to_dense(self):
        cur = self._A[0].reshape((2, self._A[0].shape[2]))
        for a in self._A[1:]:
            cur = ops.tensordot(cur, a, axes=([1], [0]))
            cur = cur.reshape((cur.shape[0] * 2, a.shape[2]))
        return cur.reshape((2**self.N,))

    def norm(self):
        self._shift_center(self._center)
        return ops.norm_fro(self._A[self._center])

    def fidelity_estimate(self):
        """Product over all splits of the kept weight fraction (``CircuitMPS.fidelity_estimate`` idea)."""
        f = 1.0
        for err in self.truncation_errors:
            f *= max(1.0 - err, 0.0)
        return f
