"""Cross-checks every array operation a caller issues on the device against numpy on the same operands
(complex-aware, fp64 reference).  ``install(mod, ...)`` patches the modules that imported ``array_contract`` /
``linalg`` by name; ``report()`` prints the mismatches."""
import numpy as np

import quimb_amd as qa
from quimb_amd import ops

bad = []
TOL = [1e-10]


def _np(x):
    x = x.to_numpy() if hasattr(x, "to_numpy") else np.asarray(x)
    return x.astype(np.complex128 if np.iscomplexobj(x) else np.float64)


def cmp(tag, got, want, shapes):
    got = _np(got)
    want = np.asarray(want)
    if got.shape != want.shape:
        bad.append((tag, shapes, "shape %s vs %s" % (got.shape, want.shape)))
        print("MISMATCH", tag, shapes, "shape", got.shape, want.shape, flush=True)
        return
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300) if want.size else 0.0
    if not err <= TOL[0]:
        bad.append((tag, shapes, float(err)))
        if len(bad) <= 15:
            print("MISMATCH", tag, shapes, "rel err %.3e" % err, flush=True)


def install(*mods, tol=1e-10):
    TOL[0] = tol
    from quimb_amd.contract import array_contract as real_ac

    def ac(arrays, inputs, output=None, **kw):
        y = real_ac(arrays, inputs, output, **kw)
        sym = {}
        eq = ",".join("".join(sym.setdefault(i, chr(97 + len(sym))) for i in t) for t in inputs)
        eq += "->" + "".join(sym[i] for i in output)
        cmp("array_contract " + eq, y, np.einsum(eq, *[_np(a) for a in arrays]), [np.shape(a) for a in arrays])
        return y

    real_td, real_mul, real_tr = ops.tensordot, ops.multiply, ops.transpose

    def td(a, b, axes=2):
        y = real_td(a, b, axes=axes)
        cmp("tensordot %s" % (axes,), y, np.tensordot(_np(a), _np(b), axes=axes), [np.shape(a), np.shape(b)])
        return y

    def mul(a, b):
        y = real_mul(a, b)
        cmp("multiply", y, _np(a) * _np(b), [np.shape(a), np.shape(b)])
        return y

    def tr(x, perm):
        y = real_tr(x, perm)
        cmp("transpose %s" % (perm,), y, np.transpose(_np(x), perm), [x.shape])
        return y

    ops.tensordot, ops.multiply, ops.transpose = td, mul, tr
    A = qa.Array
    r_gi, r_dv, r_rs, r_cj = A.__getitem__, A.__truediv__, A.reshape, A.conj

    def gi(self, key):
        y = r_gi(self, key)
        cmp("getitem %s" % (key,), y, _np(self)[key], [self.shape])
        return y

    def dv(self, other):
        y = r_dv(self, other)
        cmp("truediv", y, _np(self) / other, [self.shape])
        return y

    def rs(self, *shape):
        y = r_rs(self, *shape)
        cmp("reshape", y, _np(self).reshape(*shape), [self.shape])
        return y

    def cj(self):
        y = r_cj(self)
        cmp("conj", y, _np(self).conj(), [self.shape])
        return y

    A.__getitem__, A.__truediv__, A.reshape, A.conj = gi, dv, rs, cj

    class L:
        @staticmethod
        def qr(x):
            q, r = qa.linalg.qr(x)
            qm, rm = _np(q), _np(r)
            cmp("qr reconstruction", qm @ rm, _np(x), [x.shape])
            cmp("qr orthogonality", qm.conj().T @ qm, np.eye(qm.shape[1]), [x.shape])
            return q, r

        @staticmethod
        def svd(x):
            u, s, vh = qa.linalg.svd(x)
            um, sm, vm = _np(u), _np(s), _np(vh)
            cmp("svd reconstruction", (um * sm) @ vm, _np(x), [x.shape])
            cmp("svd values", sm, np.linalg.svd(_np(x), compute_uv=False), [x.shape])
            return u, s, vh

        @staticmethod
        def svd_via_eig(x, max_bond=-1):
            u, s, vh = qa.linalg.svd_via_eig(x, max_bond)
            um, sm, vm = _np(u), _np(s), _np(vh)
            cmp("svd_via_eig reconstruction", (um * sm) @ vm, _np(x), [x.shape])
            return u, s, vh

    for m in mods:
        if hasattr(m, "array_contract"):
            m.array_contract = ac
        if hasattr(m, "linalg"):
            m.linalg = L


def report():
    print("mismatching calls:", len(bad))
    return bad
