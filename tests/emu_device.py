"""numpy interpreter of the device-op vocabulary -- TEST INFRASTRUCTURE ONLY.

``EmuDevice`` implements exactly the operation semantics the C-ABI documents
(bundle/stride addressing of ``qamd_contract_pair``, strided ``qamd_permute``,
``qamd_reduce_sum``, ``qamd_binary`` ...), so the whole host side of quimb_amd
(planner, tree executor, slicing, exponent stripping, multi-process sharding)
can be checked against the oracle without a GPU.  It is installed with
``quimb_amd.device.set_default_device`` by the CPU tests only; the product's
default device is ``HipDevice`` and never falls back to this.
"""

import math

import numpy as np


def _offsets(groups, col):
    """Element offsets of a bundle: mixed radix over groups (last fastest)."""
    off = np.zeros(1, dtype=np.int64)
    for g in groups:
        d, s = g[0], g[col]
        off = (off[:, None] + (np.arange(d, dtype=np.int64) * s)[None, :]).reshape(-1)
    return off


class EmuDevice:
    name = "emu"

    def __init__(self):
        self.calls = {"contract_pair": 0, "permute": 0, "reduce_sum": 0, "binary": 0, "strip": 0}

    # memory
    def empty(self, n, dtype):
        return np.zeros(max(int(n), 1), dtype=np.dtype(dtype))

    def from_host(self, x):
        x = np.ascontiguousarray(x)
        return x.reshape(-1).copy() if x.size else np.zeros(1, x.dtype)

    def to_host(self, buf, n, dtype):
        return buf[: max(int(n), 0)].copy()

    def clone(self, buf):
        return buf.copy()

    def synchronize(self):
        pass

    # kernels
    def contract_pair(self, spec, dtype, a, b, c, ep=None):
        self.calls["contract_pair"] += 1
        ob_a, ob_b, ob_c = _offsets(spec.b, 1), _offsets(spec.b, 2), _offsets(spec.b, 3)
        om_a, om_c = _offsets(spec.m, 1), _offsets(spec.m, 3)
        on_b, on_c = _offsets(spec.n, 2), _offsets(spec.n, 3)
        ok_a, ok_b = _offsets(spec.k, 1), _offsets(spec.k, 2)
        A = a[ob_a[:, None, None] + om_a[None, :, None] + ok_a[None, None, :]]
        B = b[ob_b[:, None, None] + ok_b[None, :, None] + on_b[None, None, :]]
        C = np.matmul(A, B)
        if ep is not None:
            sa = float(ep[0].max()) if ep[0] is not None else 0.0
            sb = float(ep[1].max()) if ep[1] is not None else 0.0
            C = C * np.asarray(1.0 / ((sa if sa > 0 else 1.0) * (sb if sb > 0 else 1.0)), dtype=C.real.dtype)
            if ep[2] is not None and C.size:
                ep[2][self.calls["contract_pair"] % ep[2].size] = max(ep[2][self.calls["contract_pair"] % ep[2].size], np.max(np.abs(C)))
        idx = ob_c[:, None, None] + om_c[None, :, None] + on_c[None, None, :]
        assert len(np.unique(idx)) == idx.size, "output offsets collide"
        c[idx] = C

    def contract_pair_dot(self, spec, dtype, a, b, t, out, ep=None):
        """Semantics of qamd_contract_pair_dot (include/quimb_amd.h): the join's result meets ``t`` element by element and
        is summed; nothing but the scalar is written.  (The interpreter takes it for fp32 GEMM-shaped joins -- one K group,
        K >= 64 -- which is where the library's planner uses the kernel that supports it.)"""
        if np.dtype(dtype) != np.float32 or len(spec.k) != 1 or spec.K < 64:
            return False
        self.calls["pair_dot"] = self.calls.get("pair_dot", 0) + 1
        tmp = np.zeros(max(int(spec.B * spec.M * spec.N), 1), np.dtype(dtype))       # the result, C-contiguous like t
        self.contract_pair(spec, dtype, a, b, tmp, None)
        z = float(np.dot(tmp.astype(np.float64), t[: tmp.size].astype(np.float64)))
        if ep is not None:
            for sl in ep[:3]:
                m = float(sl.max()) if sl is not None else 0.0
                z /= m if m > 0 else 1.0
            if ep[3] is not None:
                ep[3][0] = abs(z)
        out[0] = z
        return True

    def contract_rowpass(self, rp, dtype, a, ws, c, ep=None):
        """Semantics of qamd_contract_rowpass (include/quimb_amd.h): five site absorptions, strided in and out."""
        self.calls["rowpass"] = self.calls.get("rowpass", 0) + 1
        D = rp.D
        if rp.s_groups is None:           # the first row: no boundary tensor, site tensors without up legs
            W = []
            for i, st in enumerate(rp.w_strides):
                dims = [(D if i else 1, st[1]), (D, st[2]), (D, st[3])]
                W.append(ws[i][_offsets(dims, 1)].reshape(D if i else 1, D, D))       # [left, down, right]
            X = np.einsum("xab,bcd,def,fgh,hij->acegij", W[0], W[1], W[2], W[3], W[4], optimize=True)   # [d1..d5, h]
            if ep is not None:
                scl = 1.0
                for t in ep[1:6]:
                    if t is not None and t.max() > 0:
                        scl *= float(t.max())
                X = X * np.asarray(1.0 / scl, dtype=X.real.dtype)
                if ep[6] is not None and X.size:
                    ep[6][0] = max(ep[6][0], np.max(np.abs(X)))
            idx = _offsets([(D, s_) for s_ in rp.sd] + [(D, rp.sh)], 1)
            assert len(np.unique(idx)) == idx.size
            c[idx] = X.reshape(-1)
            return
        os_a = _offsets([(d, sa) for d, sa, _ in rp.s_groups], 1)
        os_c = _offsets([(d, sc) for d, _, sc in rp.s_groups], 1)
        ov = _offsets([(D, s) for s in rp.sv], 1)
        T = a[os_a[:, None] + ov[None, :]].reshape((len(os_a),) + (D,) * 5)            # [S, v1..v5]
        W = []
        ed = [e or D for e in getattr(rp, "ed", (0,) * 5)]       # the new legs may be shorter than D (range-sliced cut bonds)
        eh = getattr(rp, "eh", 0) or D
        for i, st in enumerate(rp.w_strides):
            er = eh if i == 4 else D
            dims = [(D, st[0]), (D if i else 1, st[1]), (ed[i], st[2]), (er, st[3])]
            W.append(ws[i][_offsets(dims, 1)].reshape(D, D if i else 1, ed[i], er))       # [up, left, down, right]
        X = np.einsum("sabcde,axyz->sbcdeyz", T, W[0])[..., :, :]                      # site 0 (left extent 1 summed)
        # X[s, v2..v5, d1, b1] -> absorb sites 1..4
        X = np.einsum("sbcdeyz,bzpq->scdeypq", X, W[1])     # -> [s, v3, v4, v5, d1, d2, b2]
        X = np.einsum("scdeypq,cqrt->sdeyprt", X, W[2])     # -> [s, v4, v5, d1, d2, d3, b3]
        X = np.einsum("sdeyprt,dtuv->seypruv", X, W[3])     # -> [s, v5, d1, d2, d3, d4, b4]
        X = np.einsum("seypruv,evwh->sypruwh", X, W[4])     # -> [s, d1..d5, h]
        if ep is not None:
            scl = 1.0
            for t in ep[:6]:
                if t is not None and t.max() > 0:
                    scl *= float(t.max())
            X = X * np.asarray(1.0 / scl, dtype=X.real.dtype)
            if ep[6] is not None and X.size:
                ep[6][0] = max(ep[6][0], np.max(np.abs(X)))
        od = _offsets([(e, s) for e, s in zip(ed, rp.sd)] + [(eh, rp.sh)], 1)
        idx = os_c[:, None] + od[None, :]
        assert len(np.unique(idx)) == idx.size
        c[idx] = X.reshape(len(os_c), -1)

    def contract_chain2(self, c2, dtype, a, w1, w2, c, ep=None, pin=None):
        """Semantics of qamd_contract_chain2 (see include/quimb_amd.h); the small tensors arrive in
        their own layouts and are gathered as ``c2.w1_pack`` / ``w2_pack`` describe."""
        self.calls["chain2"] = self.calls.get("chain2", 0) + 1
        D = c2.D
        w1p = w1[_offsets(list(zip(c2.w1_pack.shape, c2.w1_pack.strides)), 1)]
        w2p = w2[_offsets(list(zip(c2.w2_pack.shape, c2.w2_pack.strides)), 1)]
        om_a = _offsets([(d, sa) for d, sa, _ in c2.m], 1)
        om_c = _offsets([(d, sc) for d, _, sc in c2.m], 1)
        ok1 = np.asarray(c2.off_k1, dtype=np.int64)
        ov = np.arange(D, dtype=np.int64) * c2.sa_v
        A = a[ok1[:, None, None] + ov[None, :, None] + om_a[None, None, :]]          # [k1, v, m]
        K1, NO = c2.K1, c2.NO
        W1 = w1p[: K1 * D * D].reshape(K1, D, D)                                       # [k1, x, y]
        W2 = w2p[: D * D * NO * D].reshape(D, D, NO, D)                                # [y, v, no, ni]
        X = np.einsum("kvm,kxy->xyvm", A, W1)
        Cv = np.einsum("xyvm,yvoi->omxi", X, W2)                                       # [no, m, x, ni]
        if ep is not None:
            sc = 1.0
            for t in ep[:3]:
                if t is not None and t.max() > 0:
                    sc *= float(t.max())
            Cv = Cv * np.asarray(1.0 / sc, dtype=Cv.real.dtype)
            if ep[3] is not None and Cv.size:
                ep[3][0] = max(ep[3][0], np.max(np.abs(Cv)))
        oco = np.asarray(c2.off_co, dtype=np.int64)
        idx = oco[:, None, None, None] + om_c[None, :, None, None] + (np.arange(D) * D)[None, None, :, None] + np.arange(D)[None, None, None, :]
        assert len(np.unique(idx)) == idx.size
        c[idx] = Cv

    def permute(self, dst, src, shape, strides, offset, dtype):
        self.calls["permute"] += 1
        n = int(np.prod(shape)) if len(shape) else 1
        off = np.full(1, int(offset), dtype=np.int64)
        for d, s in zip(shape, strides):
            off = (off[:, None] + (np.arange(d, dtype=np.int64) * s)[None, :]).reshape(-1)
        dst[:n] = src[off]

    def reduce_sum(self, out, x, keep_shape, keep_strides, red_shape, red_strides, dtype):
        self.calls["reduce_sum"] += 1
        ok = _offsets([(d, s) for d, s in zip(keep_shape, keep_strides)], 1)
        orr = _offsets([(d, s) for d, s in zip(red_shape, red_strides)], 1)
        out[: ok.size] = x[ok[:, None] + orr[None, :]].sum(axis=1)

    def binary(self, out, a, a_strides, b, b_strides, shape, op, dtype):
        self.calls["binary"] += 1
        oa = _offsets([(d, s) for d, s in zip(shape, a_strides)], 1)
        ob = _offsets([(d, s) for d, s in zip(shape, b_strides)], 1)
        va, vb = a[oa], b[ob]
        out[: oa.size] = {"add": np.add, "mul": np.multiply, "sub": np.subtract, "div": np.true_divide}[op](va, vb)

    def scale(self, x, n, factor, dtype):
        f = complex(factor)
        x[:n] = x[:n] * (f if np.dtype(dtype).kind == "c" else f.real)

    def axpby(self, y, x, n, fy, fx, dtype):
        y[:n] = y[:n] * fy + x[:n] * fx

    def axpby_exp(self, y, x, n, y_exp, x_exp, dtype):
        a, b = float(y_exp[0]), float(x_exp[0])
        m = max(a, b)
        fy = 10.0 ** (a - m) if np.isfinite(a) else 0.0
        fx = 10.0 ** (b - m) if np.isfinite(b) else 0.0
        y[:n] = y[:n] * np.asarray(fy, y.real.dtype) + x[:n] * np.asarray(fx, x.real.dtype)
        y_exp[0] = m

    # the vector work of a Lanczos step (csrc/krylov.hip)
    def krylov_workspace(self, rows, n, dtype):
        return np.zeros(2, np.float64)                  # [0] = ||w||^2 left by krylov_subtract

    def krylov_project(self, h, h_sum, Q, ldq, rows, w, n, accumulate, dtype, ws):
        got = np.array([np.vdot(Q[i * ldq: i * ldq + n], w[:n]) for i in range(rows)], dtype=np.dtype(dtype))
        h[:rows] = got
        if h_sum is not None:
            h_sum[:rows] = (h_sum[:rows] + got) if accumulate else got

    def krylov_subtract(self, w, Q, ldq, rows, h, n, want_norm, dtype, ws):
        for i in range(rows):
            w[:n] -= h[i] * Q[i * ldq: i * ldq + n]
        if want_norm:
            ws[0] = float(np.sum(np.abs(w[:n].astype(np.complex128 if np.dtype(dtype).kind == "c" else np.float64)) ** 2))

    def krylov_extend(self, q_next, w, n, h_j, ab, eps, dtype, ws):
        beta, alpha = float(np.sqrt(ws[0])), float(np.real(h_j[0]))
        ab[0], ab[1] = alpha, beta
        scale = 1.0 / beta if beta > eps * max(abs(alpha), 1.0) else 0.0
        q_next[:n] = (w[:n] * scale).astype(np.dtype(dtype))

    def new_exponent_neg_inf(self):
        return np.full(1, -np.inf)

    def conj(self, dst, src, n, dtype):
        dst[:n] = np.conj(src[:n])

    def cast(self, dst, dst_dtype, src, src_dtype, n):
        v = src[:n]
        if np.dtype(dst_dtype).kind != "c" and np.dtype(src_dtype).kind == "c":
            v = v.real
        dst[:n] = v.astype(dst_dtype)

    def fill(self, dst, n, value, dtype):
        dst[:n] = value if np.dtype(dtype).kind == "c" else complex(value).real

    def as_real(self, buf):
        return buf.view(buf.real.dtype)

    def complex_expand(self, dst, src, n, dtype, conj=False):
        z = np.conj(src[:n]) if conj else src[:n]
        blk = np.stack([z.real, z.imag, -z.imag, z.real], axis=1).reshape(-1)
        dst[: 4 * n] = blk

    def new_exponent(self):
        return np.zeros(1, dtype=np.float64)

    def strip_exponent(self, x, n, dtype, exponent):
        self.calls["strip"] += 1
        m = float(np.max(np.abs(x[:n]))) if n else 0.0
        if m > 0:
            x[:n] = x[:n] / np.asarray(m, dtype=x.real.dtype)
            exponent[0] += math.log10(m)

    def new_slots(self, n_tensors, dtype):
        rdt = np.float32 if np.dtype(dtype) in (np.dtype("float32"), np.dtype("complex64")) else np.float64
        return np.zeros((int(n_tensors), 64), dtype=rdt)

    def slots_row(self, slots, i):
        return slots[i]

    def slots_log10_sum(self, slots, dtype, exponent):
        m = slots.max(axis=1).astype(np.float64)
        exponent[0] += float(np.sum(np.log10(m[m > 0])))

    def div_by_absmax(self, x, n, slots_row, dtype):
        m = slots_row.max()
        if m > 0:
            x[:n] = x[:n] / m

    def read_exponent(self, exponent):
        return float(exponent[0])

    def absmax(self, x, n, dtype):
        return float(np.max(np.abs(x[:n]))) if n else 0.0

    def shares_storage(self, a, b):
        return bool(np.shares_memory(a, b))

    def buffer_address(self, buf):
        """Stand-in for a device address: a handle the interpreter can turn back into the buffer."""
        self._handles = getattr(self, "_handles", {})
        self._handles[id(buf)] = buf
        return id(buf)

    def microtree_run(self, mt, table, keep, out):
        """Semantics of qamd_microtree_run: walk the plan step by step for every instance."""
        for ii in range(table.shape[0]):
            xs = [self._handles[int(h)] for h in table[ii]]
            wdt = np.result_type(mt.dtype, np.float64) if getattr(mt, "wide", False) else mt.dtype    # QAMD_MICRO_WIDE
            arena = np.zeros(max(mt.arena_elems, 1), dtype=wdt)
            for s in mt.steps:
                sp = s["spec"]
                A = arena[s["a"][1]:] if s["a"][0] else xs[s["a"][1]]
                B = arena[s["b"][1]:] if s["b"][0] else xs[s["b"][1]]
                ob_a, ob_b, ob_c = (_offsets(sp.b, c) for c in (1, 2, 3))
                om_a, om_c = _offsets(sp.m, 1), _offsets(sp.m, 3)
                on_b, on_c = _offsets(sp.n, 2), _offsets(sp.n, 3)
                ok_a, ok_b = _offsets(sp.k, 1), _offsets(sp.k, 2)
                a4 = A[ob_a[:, None, None] + om_a[None, :, None] + ok_a[None, None, :]]      # [b, m, k]
                b4 = B[ob_b[:, None, None] + ok_b[None, :, None] + on_b[None, None, :]]      # [b, k, n]
                c = np.einsum("bmk,bkn->bmn", a4.astype(wdt), b4.astype(wdt))
                oc = ob_c[:, None, None] + om_c[None, :, None] + on_c[None, None, :]
                if s["c_off"] < 0:
                    out[ii * mt.out_elems + oc] = c
                else:
                    arena[s["c_off"] + oc] = c

    def unary(self, dst, src, n, op, dtype):
        fn = {"abs": np.abs, "sqrt": np.sqrt, "exp": np.exp, "log": np.log, "log10": np.log10}[op]
        with np.errstate(all="ignore"):
            dst[:n] = fn(src[:n])

    def minmax(self, x, n, want_min, dtype):
        out = self.empty(1, dtype)
        out[0] = np.min(x[:n]) if want_min else np.max(x[:n])
        return out

    # decompositions (quimb_amd.linalg asks the device first): numpy LAPACK on the emulated buffers
    def _wrap_np(self, x):
        from quimb_amd.array import Array

        return Array.from_numpy(np.ascontiguousarray(x), dev=self)

    def linalg_svd(self, x, full_matrices=False):
        u, s, vh = np.linalg.svd(x.to_numpy(), full_matrices=full_matrices)
        return self._wrap_np(u), self._wrap_np(s), self._wrap_np(vh)

    def linalg_qr(self, x, mode="reduced"):
        q, r = np.linalg.qr(x.to_numpy(), mode=mode)
        return self._wrap_np(q), self._wrap_np(r)

    def linalg_eigh(self, x):
        w, v = np.linalg.eigh(x.to_numpy())
        return self._wrap_np(w), self._wrap_np(v)

    def linalg_inv(self, x):
        return self._wrap_np(np.linalg.inv(x.to_numpy()))

    def linalg_pinv(self, x):
        return self._wrap_np(np.linalg.pinv(x.to_numpy()))

    def linalg_solve(self, a, b):
        return self._wrap_np(np.linalg.solve(a.to_numpy(), np.asarray(b.to_numpy() if hasattr(b, "to_numpy") else b)))

    def linalg_cholesky(self, x):
        return self._wrap_np(np.linalg.cholesky(x.to_numpy()))

    def linalg_solve_triangular(self, a, b, lower=True, left=True):
        import scipy.linalg as sla

        A, B = a.to_numpy(), np.asarray(b.to_numpy() if hasattr(b, "to_numpy") else b)
        if left:
            return self._wrap_np(sla.solve_triangular(A, B, lower=lower))
        return self._wrap_np(sla.solve_triangular(A.T, B.T, lower=not lower).T)        # X A = B  <=>  A^T X^T = B^T
