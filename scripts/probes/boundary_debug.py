"""Replays contract_boundary_2d on the device while checking EVERY contraction / product it issues against
numpy on the same operands (fp64); prints the first mismatching calls with their shapes."""
import sys

import numpy as np

sys.path.insert(0, ".")
import quimb_amd as qa
import quimb_amd.boundary as qb
from quimb_amd import ops
from oracle import np_oracle as orc

dtype = sys.argv[1] if len(sys.argv) > 1 else "float32"
bad = []
tol = 2e-4 if dtype == "float32" else 1e-10

real_ac, real_td, real_mul, real_nf = qb.array_contract, ops.tensordot, ops.multiply, ops.norm_fro


def cmp(tag, got, want, shapes):
    got = np.asarray(got.to_numpy() if hasattr(got, "to_numpy") else got, dtype=np.float64)
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)
    if not err <= tol:
        bad.append((tag, shapes, float(err)))
        if len(bad) <= 12:
            print("MISMATCH", tag, shapes, "rel err %.3e" % err, flush=True)


def ac(arrays, inputs, output):
    y = real_ac(arrays, inputs, output)
    sym = {}
    eq = ",".join("".join(sym.setdefault(i, chr(97 + len(sym))) for i in t) for t in inputs) + "->" + "".join(sym[i] for i in output)
    want = np.einsum(eq, *[a.to_numpy().astype(np.float64) for a in arrays])
    cmp("array_contract " + eq, y, want, [a.shape for a in arrays])
    return y


def td(a, b, axes):
    y = real_td(a, b, axes=axes)
    want = np.tensordot(a.to_numpy().astype(np.float64), b.to_numpy().astype(np.float64), axes=axes)
    cmp("tensordot %s" % (axes,), y, want, [a.shape, b.shape])
    return y


def mul(a, b):
    y = real_mul(a, b)
    cmp("multiply", y, a.to_numpy().astype(np.float64) * b.to_numpy().astype(np.float64), [a.shape, b.shape])
    return y


def nf(x):
    y = real_nf(x)
    want = np.linalg.norm(x.to_numpy().astype(np.float64).ravel())
    if not abs(y - want) <= tol * want:
        print("MISMATCH norm_fro", x.shape, y, want, flush=True)
        bad.append(("norm_fro", x.shape, abs(y / want - 1)))
    return y


class _Linalg:
    """qr / svd checked by reconstruction and (svd) against numpy's singular values."""

    @staticmethod
    def qr(x):
        q, r = qa.linalg.qr(x)
        xm = x.to_numpy().astype(np.float64)
        qm, rm = q.to_numpy().astype(np.float64), r.to_numpy().astype(np.float64)
        cmp("qr reconstruction", qm @ rm, xm, [x.shape])
        cmp("qr orthogonality", qm.T @ qm, np.eye(qm.shape[1]), [x.shape])
        return q, r

    @staticmethod
    def svd(x):
        u, s, vh = qa.linalg.svd(x)
        xm = x.to_numpy().astype(np.float64)
        um, sm, vm = (t.to_numpy().astype(np.float64) for t in (u, s, vh))
        if not np.abs((um * sm) @ vm - xm).max() <= tol * np.abs(xm).max() and not getattr(_Linalg, "dumped", False):
            _Linalg.dumped = True
            import torch

            np.set_printoptions(precision=6, linewidth=200)
            print("failing matrix", x.shape, "\n", repr(xm.astype(np.float32)))
            print("wrapper s", sm, "numpy s", np.linalg.svd(xm, compute_uv=False))
            xt = torch.tensor(xm.astype(np.float32), device="cuda")
            for drv in (None, "gesvd", "gesvdj"):
                u2, s2, v2 = torch.linalg.svd(xt, full_matrices=False, driver=drv)
                print(" torch driver", drv, "s", s2.cpu().numpy(), "rec err", ((u2 * s2) @ v2 - xt).abs().max().item())
            u3, s3, v3 = torch.linalg.svd(xt.double(), full_matrices=False)
            print(" torch f64 s", s3.cpu().numpy())
        cmp("svd reconstruction", (um * sm) @ vm, xm, [x.shape])
        cmp("svd values", sm, np.linalg.svd(xm, compute_uv=False), [x.shape])
        cmp("svd U orth", um.T @ um, np.eye(um.shape[1]), [x.shape])
        cmp("svd V orth", vm @ vm.T, np.eye(vm.shape[0]), [x.shape])
        return u, s, vh


Array = qa.Array
real_getitem, real_div, real_reshape = Array.__getitem__, Array.__truediv__, Array.reshape


def gi(self, key):
    y = real_getitem(self, key)
    cmp("getitem %s" % (key,), y, self.to_numpy().astype(np.float64)[key], [self.shape])
    return y


def dv(self, other):
    y = real_div(self, other)
    cmp("truediv", y, self.to_numpy().astype(np.float64) / other, [self.shape])
    return y


def rs(self, *shape):
    y = real_reshape(self, *shape)
    cmp("reshape", y, self.to_numpy().astype(np.float64).reshape(*shape), [self.shape])
    return y


Array.__getitem__, Array.__truediv__, Array.reshape = gi, dv, rs
real_tr = ops.transpose


def tr(x, perm):
    y = real_tr(x, perm)
    cmp("transpose %s" % (perm,), y, np.transpose(x.to_numpy().astype(np.float64), perm), [x.shape])
    return y


ops.transpose = tr
qb.linalg = _Linalg
qb.array_contract = ac
ops.tensordot = lambda a, b, axes=2: td(a, b, axes)
ops.multiply = mul
ops.norm_fro = nf

L = int(sys.argv[2]) if len(sys.argv) > 2 else 16
arrays, _ = orc.tn2d_classical_ising(L, L, 0.44)
m, e = qa.contract_boundary_2d(arrays, L, L, max_bond=8, strip_exponent=True, dtype=dtype)
mo, eo = orc.oracle_contract_boundary_2d(arrays, L, L, max_bond=8)
print("device", m, e, "oracle", mo / abs(mo), eo + np.log10(abs(mo)), "mismatching calls:", len(bad))
