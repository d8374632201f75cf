"""Device layer: the only module that touches ``libquimb_amd.so`` and torch.

``HipDevice`` owns nothing but caches: device memory comes from torch's ROCm
allocator (plumbing, per the design brief), every arithmetic / layout operation
is a call through the C-ABI in ``include/quimb_amd.h`` on torch's current HIP
stream.  There is deliberately no CPU implementation here -- constructing the
default device without the HIP library or without a GPU raises.
"""

import ctypes as C
import os

import contextlib
import threading

import numpy as np

from . import _lib
from .options import get_options
from .pairwise import GettSpec, prod

_DT_CODE = {
    np.dtype("float32"): _lib.QAMD_F32,
    np.dtype("float64"): _lib.QAMD_F64,
    np.dtype("complex64"): _lib.QAMD_C64,
    np.dtype("complex128"): _lib.QAMD_C128,
}


def dtype_code(dtype):
    try:
        return _DT_CODE[np.dtype(dtype)]
    except KeyError:
        raise TypeError(f"quimb_amd supports float32/float64/complex64/complex128, got {dtype}") from None


def _i64arr(xs):
    xs = [int(x) for x in xs]
    return (C.c_int64 * max(len(xs), 1))(*xs)


def fill_plan_struct(spec: GettSpec, code: int):
    """GettSpec -> ``qamd_pair_plan`` (unfinalised)."""
    p = _lib.PairPlanStruct()
    p.dtype = code
    p.nb, p.nm, p.nn, p.nk = len(spec.b), len(spec.m), len(spec.n), len(spec.k)
    for i, (d, sa, sb, sc) in enumerate(spec.b):
        p.dim_b[i], p.sa_b[i], p.sb_b[i], p.sc_b[i] = d, sa, sb, sc
    for i, (d, sa, _, sc) in enumerate(spec.m):
        p.dim_m[i], p.sa_m[i], p.sc_m[i] = d, sa, sc
    for i, (d, _, sb, sc) in enumerate(spec.n):
        p.dim_n[i], p.sb_n[i], p.sc_n[i] = d, sb, sc
    for i, (d, sa, sb, _) in enumerate(spec.k):
        p.dim_k[i], p.sa_k[i], p.sb_k[i] = d, sa, sb
    p.tile_cfg = -1
    p.split_k = 0
    p.kernel = 0
    return p


class _CompiledPair:
    __slots__ = ("struct", "ktab", "ws_bytes", "ready", "ready_stream")


class HipDevice:
    """MI355X device: torch-ROCm memory + hand-written HIP kernels via ctypes."""

    name = "hip"

    @property
    def record(self):
        """The launch-program recorder of THIS thread (or None).  The C recorder is thread-local (csrc/program.cpp: ``g_rec``),
        so the Python side must be too: while one thread records, the launches and allocations of every other thread
        using this device go to the GPU and the caching allocator as usual -- not into the recorder's pool."""
        tls = self.__dict__.get("_rec_tls")
        return getattr(tls, "rec", None) if tls is not None else None

    @record.setter
    def record(self, rec):
        tls = self.__dict__.get("_rec_tls")
        if tls is None:                      # (a subclass that skips __init__: the record-only device of the CPU tests)
            tls = self.__dict__["_rec_tls"] = threading.local()
        tls.rec = rec

    def __init__(self, index=None):
        self.lib = _lib.load()
        import torch

        self.torch = torch
        if not torch.cuda.is_available():
            raise _lib.QamdError(
                "quimb_amd: no HIP device visible (torch.cuda.is_available() is False); "
                "the contraction backend has no CPU fallback"
            )
        if index is None:
            index = torch.cuda.current_device()
        self.index = index
        self.tdev = torch.device("cuda", index)
        self._tdt = {
            np.dtype("float32"): torch.float32,
            np.dtype("float64"): torch.float64,
            np.dtype("complex64"): torch.complex64,
            np.dtype("complex128"): torch.complex128,
            np.dtype("int64"): torch.int64,
        }
        self._pairs = {}
        #: the active launch-program recorder (quimb_amd/program.py) or None: while set, every allocation of this
        #: device comes from the recorder's pool and the library appends launches to its program instead of issuing them
        self._rec_tls = threading.local()
        self.record = None
        #: set to a list to collect (spec, dtype, tile_cfg, split_k, start_event, end_event)
        #: per qamd_contract_pair launch (HIP events on the launch stream)
        self.profile = None
        #: launches below this many multiplications are not bracketed by events when ``profile`` is a list
        self.profile_min_mults = 0
        # Kernel pins (quimb_amd/options.py: pair_kernel / tile_cfg / split_k / micro_arena).  Who decides, in this order:
        # (1) the executor / expression that is running on this thread -- it installs the options it CAPTURED WHEN BUILT for
        # the duration of its run (``pinned``); (2) these attributes when a script set them (None = not set); (3) the
        # thread's current options at the time of the call (a bare ``qa.tensordot`` under ``with qa.options(...)``).
        self.force_tile_cfg = None
        self.force_split_k = None
        #: 0 = auto (streaming kernel where eligible), -1 = always the tiled GETT kernel
        self.force_kernel = None
        #: fused-pair kernel pin: "auto" | "lds" | "reg" | "quad" (QAMD_CHAIN2_FORCE_* flag bits of the plan)
        self.force_chain2 = get_options().chain2_kernel
        self.micro_arena = None

    def _pins(self):
        """(pair_kernel, tile_cfg, split_k, micro_arena) in force for THIS call (see ``__init__``)."""
        o = getattr(self._rec_tls, "pins", None) or get_options()
        pick = lambda dev_value, opt_value: opt_value if dev_value is None else dev_value
        kernel = int(pick(self.force_kernel, o.pair_kernel))
        if kernel == 0 and o.join_arith != "f32":
            # (opt-in: the planner puts the pairs large enough for it on gemmh.hip -- the k-outer joins, or with "-all" any
            # operand layout --, everything else as usual)
            kernel = -7 if o.join_arith == "f16x3" else -8
        return (kernel, int(pick(self.force_tile_cfg, o.tile_cfg)),
                int(pick(self.force_split_k, o.split_k)), pick(self.micro_arena, o.micro_arena))

    @contextlib.contextmanager
    def pinned(self, options):
        """The kernel pins of ``options`` (what an executor captured when it was built) for the calls this thread makes
        inside the block."""
        old = getattr(self._rec_tls, "pins", None)
        self._rec_tls.pins = options
        try:
            yield
        finally:
            self._rec_tls.pins = old

    # ---- memory ---------------------------------------------------------
    def empty(self, n, dtype):
        if self.record is not None:
            return self.record.alloc(max(int(n), 1), self._tdt[np.dtype(dtype)])
        return self.torch.empty(max(int(n), 1), dtype=self._tdt[np.dtype(dtype)], device=self.tdev)

    def _zeros(self, n, tdtype, value=0.0):
        """``n`` elements of a torch dtype holding ``value``; recorded as a fill while a program is being recorded (so
        that every replay starts from it), a plain torch allocation otherwise."""
        if self.record is None:
            return self.torch.full((int(n),), value, dtype=tdtype, device=self.tdev)
        t = self.record.alloc(int(n), tdtype)
        code = {self.torch.float32: _lib.QAMD_F32, self.torch.float64: _lib.QAMD_F64}[tdtype]
        _lib.check(self.lib.qamd_fill(t.data_ptr(), int(n), float(value), 0.0, code, self.stream()), "qamd_fill")
        return t

    def from_host(self, x):
        x = np.ascontiguousarray(x)
        t = self.torch.from_numpy(x.reshape(-1) if x.size else np.zeros(1, x.dtype))
        return t.to(self.tdev)

    def to_host(self, buf, n, dtype):
        return buf[: max(int(n), 0)].cpu().numpy().astype(np.dtype(dtype), copy=False)

    def clone(self, buf):
        return buf.clone()

    def ptr(self, buf):
        return buf.data_ptr()

    def stream(self):
        return self.torch.cuda.current_stream(self.tdev).cuda_stream

    def synchronize(self):
        self.torch.cuda.synchronize(self.tdev)

    def lane_streams(self, n, priorities=None, own_lane0=False):
        """[current stream, side stream 1, ...]: the HIP streams the tree executor runs independent branches on.
        ``priorities``: per lane, -1 = high, 0 = normal (HIP stream priorities: the dispatcher serves a high-priority
        queue first when several have work).  ``own_lane0``: lane 0 is a stream of the pool too (a launch program forks
        from and joins back to the caller's stream around its run) instead of the caller's current stream."""
        pool = getattr(self, "_lane_pool", None)
        if pool is None:
            pool = self._lane_pool = {}
        out = []
        for lane in range(n):
            if lane == 0 and not own_lane0:
                out.append(self.torch.cuda.current_stream(self.tdev))
                continue
            pr = int(priorities[lane]) if priorities is not None else 0
            key = (lane, pr)
            st = pool.get(key)
            if st is None:
                st = pool[key] = self.torch.cuda.Stream(device=self.tdev, priority=pr)
            out.append(st)
        return out

    def _workspace(self, nbytes):
        """Split-K / dot workspace of ONE launch: a stream-ordered allocation from torch's caching allocator on the
        launch stream.  The executor runs independent branches on several HIP streams (lanes), so there is no
        device-wide workspace: two under-filled launches on different lanes must not write their partial sums into
        the same slab.  The block goes back to its stream's pool as soon as the launch is queued (later launches on
        that stream are ordered behind it); under stream capture it comes from the graph's private pool and lives as
        long as the graph does."""
        if nbytes <= 0:
            return None, None, 0
        ws = (self.record.alloc(int(nbytes), self.torch.uint8) if self.record is not None else
              self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.tdev))
        return ws, ws.data_ptr(), ws.numel()

    def release_temp(self, t):
        """A temporary of ONE launch (workspace, scratch, expanded operand) is done with: a no-op in ordinary operation
        (the caching allocator sees the tensor die), the hand-back to the pool while a program is being recorded."""
        if t is not None and self.record is not None:
            self.record.release(t)

    def _scratch(self):
        """4 doubles of reduction scratch for one call, private to the launch stream (see ``_workspace``)."""
        if self.record is not None:
            return self.record.alloc(4, self.torch.float64)
        return self.torch.empty(4, dtype=self.torch.float64, device=self.tdev)

    # ---- pairwise contraction ---------------------------------------------
    def compile_pair(self, spec, dtype, align_a=16, align_b=16, align_c=16):
        code = dtype_code(dtype)
        kernel, tile_cfg, split_k, _ = self._pins()
        key = (spec, code, align_a, align_b, tile_cfg, split_k, align_c, kernel)
        cp = self._pairs.get(key)
        if cp is not None:
            return cp
        p = fill_plan_struct(spec, code)
        p.tile_cfg = tile_cfg
        p.split_k = split_k
        p.kernel = kernel
        _lib.check(
            self.lib.qamd_pair_plan_finalize(C.byref(p), align_a, align_b, align_c), "qamd_pair_plan_finalize"
        )
        klen = self.lib.qamd_pair_ktab_len(C.byref(p))
        ktab = self.torch.empty(int(klen), dtype=self.torch.int64, device=self.tdev)
        _lib.check(
            self.lib.qamd_pair_build_ktab(C.byref(p), ktab.data_ptr(), self.stream()), "qamd_pair_build_ktab"
        )
        cp = _CompiledPair()
        cp.struct = p
        cp.ktab = ktab
        # the table is filled by a kernel on THIS stream; a launch on another lane that finds the plan in the cache
        # waits for the event first (``_wait_plan_tables``)
        cp.ready = self.torch.cuda.Event()
        cp.ready.record()
        cp.ready_stream = self.stream()
        cp.ws_bytes = int(self.lib.qamd_pair_workspace_bytes(C.byref(p)))
        self._pairs[key] = cp
        return cp

    def _wait_plan_tables(self, cp):
        """Order the current stream behind the kernel that filled ``cp.ktab`` (built on another stream)."""
        ev = cp.ready
        if ev is None or self.record is not None:     # (a recorder synchronises once, after the recording)
            return
        if self.stream() != cp.ready_stream:
            self.torch.cuda.current_stream(self.tdev).wait_event(ev)
        if not self.torch.cuda.is_current_stream_capturing() and ev.query():
            cp.ready = None          # the table is final: nothing to wait for any more

    def describe_pair(self, cp):
        buf = C.create_string_buffer(160)
        self.lib.qamd_pair_describe(C.byref(cp.struct), buf, 160)
        return buf.value.decode()

    def contract_pair(self, spec, dtype, a, b, c, ep=None):
        """C = A . B.  ``ep`` = (slots_a, slots_b, slots_out) enables the fused
        exponent-stripping epilogue (entries may be None)."""
        pa, pb, pc = a.data_ptr(), b.data_ptr(), c.data_ptr()
        cp = self.compile_pair(spec, dtype, min(pa & -pa, 16), min(pb & -pb, 16), min(pc & -pc, 16))
        ws_keep, ws, wsn = self._workspace(cp.ws_bytes)
        self._wait_plan_tables(cp)
        epp = None
        if ep is not None:
            e = _lib.Epilogue()
            e.scale_a = ep[0].data_ptr() if ep[0] is not None else None
            e.scale_b = ep[1].data_ptr() if ep[1] is not None else None
            e.absmax_out = ep[2].data_ptr() if ep[2] is not None else None
            epp = C.byref(e)
        prof = self.profile
        if prof is not None and getattr(spec, "mults", self.profile_min_mults) < self.profile_min_mults:
            prof = None      # an event pair costs a small launch ~10 us of queue time: only the launches that matter
        if self.record is not None:
            prof = None
            self.record.maybe_mark(spec, np.dtype(dtype), lambda: (self.describe_pair(cp), cp.struct.split_k))
        if prof is not None:
            e0 = self.torch.cuda.Event(enable_timing=True)
            e1 = self.torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(
            self.lib.qamd_contract_pair_ex(
                C.byref(cp.struct), pa, pb, pc, cp.ktab.data_ptr(), ws, wsn, epp, self.stream()
            ),
            "qamd_contract_pair_ex",
        )
        if prof is not None:
            e1.record()
            prof.append((spec, np.dtype(dtype), self.describe_pair(cp), cp.struct.split_k, e0, e1))
        self.release_temp(ws_keep)

    def contract_pair_dot(self, spec, dtype, a, b, t, out, ep=None):
        """``out[0] = sum((A . B) * T)`` with the product never stored (gemmk.hip, DOT variant): a join whose result is
        only consumed by one inner product with ``t``, a tensor of the layout the result would have had.  ``ep`` =
        (slots_a, slots_b, slots_t, slots_out) or None.  Returns False -- nothing launched -- when the planner did not put
        this contraction on the kernel that can do it; the caller then issues the two steps."""
        if np.dtype(dtype) != np.dtype("float32"):
            return False
        pa, pb, pt = a.data_ptr(), b.data_ptr(), t.data_ptr()
        cp = self.compile_pair(spec, dtype, min(pa & -pa, 16), min(pb & -pb, 16), min(pt & -pt, 16))
        if cp.struct.kernel not in (5, 7):
            return False
        nws = int(self.lib.qamd_pair_dot_workspace_bytes(C.byref(cp.struct)))
        if nws <= 0:
            return False
        ws_keep, ws, wsn = self._workspace(nws)
        ptr = lambda x: (x.data_ptr() if x is not None else None)
        epp, st = None, None
        if ep is not None:
            e = _lib.Epilogue()
            e.scale_a, e.scale_b, e.absmax_out = ptr(ep[0]), ptr(ep[1]), ptr(ep[3])
            epp, st = C.byref(e), ptr(ep[2])
        prof = self.profile
        if prof is not None and getattr(spec, "mults", self.profile_min_mults) < self.profile_min_mults:
            prof = None
        name = lambda: self.describe_pair(cp) + " + dot"
        if self.record is not None:
            prof = None
            self.record.maybe_mark(spec, np.dtype(dtype), lambda: (name(), 1))
        if prof is not None:
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(self.lib.qamd_contract_pair_dot(C.byref(cp.struct), pa, pb, pt, out.data_ptr(), ws, wsn, epp, st,
                                                   self.stream()), "qamd_contract_pair_dot")
        if prof is not None:
            e1.record()
            prof.append((spec, np.dtype(dtype), name(), 1, e0, e1))
        self.release_temp(ws_keep)
        return True

    # ---- one row of a boundary sweep in one launch --------------------------------------
    def contract_rowpass(self, rp, dtype, a, ws, c, ep=None):
        """C = the five site absorptions of one row applied to A (rowpass.hip).  ``rp``: pairwise.RowpassSpec; ``ws``:
        the five site tensors in their own layouts; ``ep`` = (slots_a, slots_w0 .. slots_w4, slots_out) or None."""
        key = ("rowpass", rp, dtype_code(dtype))
        pl = self._pairs.get(key)
        if pl is None:
            pl = _lib.RowpassPlanStruct()
            pl.dtype, pl.D, pl.nsites = dtype_code(dtype), rp.D, len(rp.sv)
            pl.nS = -1 if rp.s_groups is None else len(rp.s_groups)          # -1: the first row, no boundary tensor
            for i in range(5):
                pl.sv[i], pl.sd[i] = rp.sv[i], rp.sd[i]
                for j in range(4):
                    pl.w_strides[i][j] = rp.w_strides[i][j]
            pl.sh = rp.sh
            for i, (d, sa_, sc_) in enumerate(rp.s_groups or ()):
                pl.dim_s[i], pl.sa_s[i], pl.sc_s[i] = d, sa_, sc_
            for i in range(5):
                pl.ed[i] = rp.ed[i]
            pl.eh, pl.kernel = rp.eh, rp.kernel
            self._pairs[key] = pl
        ptr = lambda t: (t.data_ptr() if t is not None else None)
        wp = (C.c_void_p * 5)(*[w.data_ptr() for w in ws])
        sa = so = None
        sw = None
        if ep is not None:
            sa, so = ptr(ep[0]), ptr(ep[6])
            sw = (C.c_void_p * 5)(*[ptr(t) for t in ep[1:6]])
        name = "rowfirst_kernel" if rp.s_groups is None else (
            f"rowpass_kernel<{rp.D}, {len(rp.sv)}>" if rp.kernel == 1 else
            f"rowq_kernel<{'true' if not any(rp.ed) and not rp.eh else 'false'}>")
        prof = self.profile
        if prof is not None and rp.mults < self.profile_min_mults:
            prof = None
        if self.record is not None:
            prof = None
            self.record.maybe_mark(rp, np.dtype(dtype), lambda: (name, 1))
        if prof is not None:
            e0 = self.torch.cuda.Event(enable_timing=True)
            e1 = self.torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(self.lib.qamd_contract_rowpass(C.byref(pl), None if a is None else a.data_ptr(), wp, c.data_ptr(), sa, sw, so,
                                                  self.stream()),
                   "qamd_contract_rowpass")
        if prof is not None:
            e1.record()
            prof.append((rp, np.dtype(dtype), name, 1, e0, e1))

    # ---- fused pair of streaming steps ---------------------------------------------
    def contract_chain2(self, c2, dtype, a, w1, w2, c, ep=None, pin=None):
        """C = (A . W1) . W2 in one pass (chain2r.hip / chain2.hip).  ``c2``: pairwise.Chain2Spec;
        ``w1`` / ``w2``: the small tensors in their own layouts (``c2.w1_pack`` / ``w2_pack`` say how
        to address them); ``ep`` = (slots_a, slots_w1, slots_w2, slots_out) or None."""
        # the kernel choice (and with it the W addressing mode) follows the caller's pin (an executor's
        # ``options.chain2_kernel``), else the device's
        force = {"auto": 0, "lds": 16, "reg": 32, "quad": 64}[pin if pin not in (None, "auto") else self.force_chain2]
        key = ("chain2", c2, dtype_code(dtype), force)
        ent = self._pairs.get(key)
        if ent is None:
            pl = _lib.Chain2PlanStruct()
            pl.dtype, pl.D, pl.nm = dtype_code(dtype), c2.D, len(c2.m)
            for i, (d, sa, sc) in enumerate(c2.m):
                pl.dim_m[i], pl.sa_m[i], pl.sc_m[i] = d, sa, sc
            pl.sa_v = c2.sa_v
            if all(o % 4 == 0 for o in c2.off_co) and all(sc % 4 == 0 for (_, _, sc) in c2.m[:-1]):
                pl.flags = 1  # QAMD_CHAIN2_C_ALIGNED16
            pl.flags |= force
            if c2.k1_single:
                pl.flags |= 2  # QAMD_CHAIN2_K1_SINGLE
            if c2.no_n2out:
                pl.flags |= 4  # QAMD_CHAIN2_NO_N2OUT
            buf = C.create_string_buffer(128)
            _lib.check(self.lib.qamd_chain2_describe(C.byref(pl), buf, 128), "qamd_chain2_describe")
            name = buf.value.decode()
            if name.startswith(("chain2r", "chain2q")):
                # the register kernel reads the small tensors in place: no packed copies, no permute launches
                pl.flags |= 8  # QAMD_CHAIN2_W_STRIDED
                s1, s2 = list(c2.w1_pack.strides), list(c2.w2_pack.strides)
                if c2.k1_single:
                    s1 = [s1[0], 0] + s1[1:]
                if c2.no_n2out:
                    s2 = s2[:2] + [0] + s2[2:]
                for i in range(4):
                    pl.w1_strides[i], pl.w2_strides[i] = s1[i], s2[i]
            k1 = self.torch.tensor(c2.off_k1, dtype=self.torch.int64, device=self.tdev)
            co = self.torch.tensor(c2.off_co, dtype=self.torch.int64, device=self.tdev)
            ent = (pl, k1, co, name)
            self._pairs[key] = ent
        pl, k1, co, name = ent
        if pl.flags & 8:
            w1p, w2p = w1, w2
        else:
            w1p = self.empty(prod(c2.w1_pack.shape), dtype)
            w2p = self.empty(prod(c2.w2_pack.shape), dtype)
            self.permute(w1p, w1, c2.w1_pack.shape, c2.w1_pack.strides, 0, dtype)
            self.permute(w2p, w2, c2.w2_pack.shape, c2.w2_pack.strides, 0, dtype)
        ptr = lambda t: (t.data_ptr() if t is not None else None)
        sa = s1 = s2 = so = None
        if ep is not None:
            sa, s1, s2, so = (ptr(t) for t in ep)
        prof = self.profile
        if prof is not None and c2.mults < self.profile_min_mults:
            prof = None
        if self.record is not None:
            prof = None
            self.record.maybe_mark(c2, np.dtype(dtype), lambda: (name, 1))
        if prof is not None:
            e0 = self.torch.cuda.Event(enable_timing=True)
            e1 = self.torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(
            self.lib.qamd_contract_chain2(
                C.byref(pl), a.data_ptr(), w1p.data_ptr(), w2p.data_ptr(), c.data_ptr(), k1.data_ptr(), co.data_ptr(),
                sa, s1, s2, so, self.stream(),
            ),
            "qamd_contract_chain2",
        )
        if prof is not None:
            e1.record()
            prof.append((c2, np.dtype(dtype), name, 1, e0, e1))
        if not (pl.flags & 8):
            self.release_temp(w1p)
            self.release_temp(w2p)

    # ---- fused triple of streaming steps --------------------------------------------
    # ---- layout / elementwise -----------------------------------------------
    def permute(self, dst, src, shape, strides, offset, dtype):
        nd = len(shape)
        _lib.check(
            self.lib.qamd_permute(
                dst.data_ptr(), src.data_ptr(), nd, _i64arr(shape), _i64arr(strides), int(offset),
                dtype_code(dtype), self.stream(),
            ),
            "qamd_permute",
        )

    def reduce_sum(self, out, x, keep_shape, keep_strides, red_shape, red_strides, dtype):
        _lib.check(
            self.lib.qamd_reduce_sum(
                out.data_ptr(), x.data_ptr(), len(keep_shape), _i64arr(keep_shape), _i64arr(keep_strides),
                len(red_shape), _i64arr(red_shape), _i64arr(red_strides), dtype_code(dtype), self.stream(),
            ),
            "qamd_reduce_sum",
        )

    def binary(self, out, a, a_strides, b, b_strides, shape, op, dtype):
        _lib.check(
            self.lib.qamd_binary(
                out.data_ptr(), a.data_ptr(), _i64arr(a_strides), b.data_ptr(), _i64arr(b_strides),
                len(shape), _i64arr(shape), {"add": 0, "mul": 1, "sub": 2, "div": 3}[op], dtype_code(dtype), self.stream(),
            ),
            "qamd_binary",
        )

    def scale(self, x, n, factor, dtype):
        f = complex(factor)
        _lib.check(self.lib.qamd_scale(x.data_ptr(), int(n), f.real, f.imag, dtype_code(dtype), self.stream()), "qamd_scale")

    def axpby(self, y, x, n, fy, fx, dtype):
        _lib.check(
            self.lib.qamd_axpby(y.data_ptr(), x.data_ptr(), int(n), float(fy), float(fx), dtype_code(dtype), self.stream()),
            "qamd_axpby",
        )

    def axpby_exp(self, y, x, n, y_exp, x_exp, dtype):
        """y*10^y_exp + x*10^x_exp -> y*10^max(...) with both exponents on the device; y_exp is advanced."""
        _lib.check(
            self.lib.qamd_axpby_exp(y.data_ptr(), x.data_ptr(), int(n), y_exp.data_ptr(), x_exp.data_ptr(),
                                    dtype_code(dtype), self.stream()),
            "qamd_axpby_exp",
        )

    def new_exponent_neg_inf(self):
        return self._zeros(1, self.torch.float64, float("-inf"))

    def conj(self, dst, src, n, dtype):
        _lib.check(self.lib.qamd_conj(dst.data_ptr(), src.data_ptr(), int(n), dtype_code(dtype), self.stream()), "qamd_conj")

    def cast(self, dst, dst_dtype, src, src_dtype, n):
        _lib.check(
            self.lib.qamd_cast(dst.data_ptr(), dtype_code(dst_dtype), src.data_ptr(), dtype_code(src_dtype), int(n), self.stream()),
            "qamd_cast",
        )

    def fill(self, dst, n, value, dtype):
        v = complex(value)
        _lib.check(self.lib.qamd_fill(dst.data_ptr(), int(n), v.real, v.imag, dtype_code(dtype), self.stream()), "qamd_fill")

    # ---- the vector work of a Lanczos step (krylov.hip) ---------------------------------
    def krylov_workspace(self, rows, n, dtype):
        nb = int(self.lib.qamd_krylov_workspace_bytes(int(rows), int(n), dtype_code(dtype)))
        if nb < 0:
            raise _lib.QamdError("qamd_krylov_workspace_bytes failed")
        return self.torch.empty(nb, dtype=self.torch.uint8, device=self.tdev)

    def krylov_project(self, h, h_sum, Q, ldq, rows, w, n, accumulate, dtype, ws):
        """h[i] = <Q_i, w>, i < rows; ``h_sum`` (or None): = h, or += h when ``accumulate``"""
        _lib.check(self.lib.qamd_krylov_project(h.data_ptr(), None if h_sum is None else h_sum.data_ptr(), Q.data_ptr(),
                                                int(ldq), int(rows), w.data_ptr(), int(n), int(bool(accumulate)),
                                                dtype_code(dtype), ws.data_ptr(), self.stream()),
                   "qamd_krylov_project")

    def krylov_subtract(self, w, Q, ldq, rows, h, n, want_norm, dtype, ws):
        """w -= sum_i h[i] Q_i (in place); ``want_norm``: leave ||w||^2 for ``krylov_extend`` in ``ws``"""
        _lib.check(self.lib.qamd_krylov_subtract(w.data_ptr(), Q.data_ptr(), int(ldq), int(rows), h.data_ptr(), int(n),
                                                 int(bool(want_norm)), dtype_code(dtype), ws.data_ptr(), self.stream()),
                   "qamd_krylov_subtract")

    def krylov_extend(self, q_next, w, n, h_j, ab, eps, dtype, ws):
        """ab[0:2] = (Re h_j[0], ||w||); q_next = w / ||w|| (zero on breakdown)"""
        _lib.check(self.lib.qamd_krylov_extend(q_next.data_ptr(), w.data_ptr(), int(n), h_j.data_ptr(), ab.data_ptr(),
                                               float(eps), dtype_code(dtype), ws.data_ptr(), self.stream()),
                   "qamd_krylov_extend")

    # ---- complex support ----------------------------------------------------------
    def as_real(self, buf):
        """Interleaved (re, im) real view of a complex buffer (shares memory)."""
        return self.torch.view_as_real(buf).reshape(-1)

    def complex_expand(self, dst, src, n, dtype, conj=False):
        _lib.check(
            self.lib.qamd_complex_expand(dst.data_ptr(), src.data_ptr(), int(n), int(bool(conj)), dtype_code(dtype), self.stream()),
            "qamd_complex_expand",
        )

    # ---- exponent stripping ---------------------------------------------------
    def new_exponent(self):
        """Device-resident float64 accumulator for log10 factors."""
        return self._zeros(1, self.torch.float64)

    def strip_exponent(self, x, n, dtype, exponent):
        """x /= max|x|; exponent += log10(max|x|) (an atomic add: lanes share one accumulator)."""
        scratch = self._scratch()
        _lib.check(
            self.lib.qamd_strip_exponent(
                x.data_ptr(), int(n), dtype_code(dtype), scratch.data_ptr(), exponent.data_ptr(), self.stream()
            ),
            "qamd_strip_exponent",
        )
        self.release_temp(scratch)

    # fused form: per-tensor absmax slots, consumed as scales by the next contraction
    def new_slots(self, n_tensors, dtype):
        """(n_tensors, 64) zeroed slots of the real dtype matching ``dtype``."""
        rdt = self.torch.float32 if np.dtype(dtype) in (np.dtype("float32"), np.dtype("complex64")) else self.torch.float64
        return self._zeros(int(n_tensors) * _lib.ABSMAX_SLOTS, rdt).reshape(int(n_tensors), _lib.ABSMAX_SLOTS)

    def slots_row(self, slots, i):
        return slots[i]

    def slots_log10_sum(self, slots, dtype, exponent):
        """exponent[0] += sum_t log10(max(slots[t]))  (device side, one launch, no sync)."""
        _lib.check(
            self.lib.qamd_absmax_log10_sum_add(slots.data_ptr(), slots.shape[0], dtype_code(dtype), exponent.data_ptr(),
                                               self.stream()),
            "qamd_absmax_log10_sum_add",
        )

    def div_by_absmax(self, x, n, slots_row, dtype):
        _lib.check(
            self.lib.qamd_div_by_absmax(x.data_ptr(), int(n), slots_row.data_ptr(), dtype_code(dtype), self.stream()),
            "qamd_div_by_absmax",
        )

    def read_exponent(self, exponent):
        if self.record is not None:
            raise _lib.QamdError("read_exponent() reads the device: not available while a launch program is being recorded")
        return float(exponent.cpu()[0])

    def absmax(self, x, n, dtype):
        if self.record is not None:
            raise _lib.QamdError("absmax() reads the device: not available while a launch program is being recorded")
        scratch = self._scratch()
        _lib.check(
            self.lib.qamd_absmax(scratch.data_ptr() + 16, x.data_ptr(), int(n), dtype_code(dtype), self.stream()),
            "qamd_absmax",
        )
        return float(scratch.cpu()[2])


    def buffer_address(self, buf):
        return buf.data_ptr()

    def is_capturing(self):
        """Is the current stream inside a hipGraph capture?"""
        try:
            return bool(self.torch.cuda.is_current_stream_capturing())
        except Exception:
            return False

    def shares_storage(self, a, b):
        """Do two buffers (views included) live in the same allocation?"""
        return a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()

    def microtree_run(self, mt, table, keep, out):
        """One launch for ``table.shape[0]`` instances of the compiled tree ``mt`` (microtree.hip).
        ``table``: int64 [ninst, ninputs] device addresses; ``keep``: objects that own those buffers."""
        torch = self.torch
        plan = getattr(mt, "_plan_dev", None)           # the plan lives (and dies) with its MicroTree
        if plan is None:
            sb, etab, ktab = mt.packed()
            plan = mt._plan_dev = (
                torch.from_numpy(np.frombuffer(sb, dtype=np.uint8).copy()).to(self.tdev),
                torch.from_numpy(etab).to(self.tdev),
                torch.from_numpy(ktab).to(self.tdev),
            )
        steps_dev, etab_dev, ktab_dev = plan
        ninst = int(table.shape[0])
        ptrs = torch.from_numpy(np.ascontiguousarray(table, dtype=np.int64)).to(self.tdev, non_blocking=False)
        mode = self._pins()[3]                            # auto | lds | global
        use_lds = mt.lds_ok and (mode == "lds" or (mode == "auto" and ninst <= 2 * 256))
        wide = bool(getattr(mt, "wide", False))
        wdt = {np.dtype("float32"): np.dtype("float64"), np.dtype("complex64"): np.dtype("complex128")}.get(np.dtype(mt.dtype), mt.dtype) \
            if wide else mt.dtype
        arena = None if use_lds else self.empty(max(mt.arena_elems * ninst, 1), wdt)
        _lib.check(
            self.lib.qamd_microtree_run_ex(
                dtype_code(mt.dtype), steps_dev.data_ptr(), len(mt.steps), etab_dev.data_ptr(), ktab_dev.data_ptr(),
                ptrs.data_ptr(), mt.ninputs,
                None if use_lds else arena.data_ptr(), int(mt.arena_elems), out.data_ptr(), int(mt.out_elems), ninst,
                1 if wide else 0, self.stream(),
            ),
            "qamd_microtree_run_ex",
        )
        # the pointer table, the arena and every input must outlive the asynchronous launch
        self._micro_keep = (ptrs, arena, keep)

    _UNARY = {"abs": 0, "sqrt": 1, "exp": 2, "log": 3, "log10": 4}

    def unary(self, dst, src, n, op, dtype):
        """dst[i] = op(src[i]); complex dtypes support only ``abs`` (dst real)."""
        _lib.check(
            self.lib.qamd_unary(dst.data_ptr(), src.data_ptr(), int(n), self._UNARY[op], dtype_code(dtype), self.stream()),
            "qamd_unary",
        )

    def minmax(self, x, n, want_min, dtype):
        """1-element device buffer holding max (or min) of a real array."""
        out = self.empty(1, dtype)
        _lib.check(
            self.lib.qamd_minmax(out.data_ptr(), x.data_ptr(), int(n), 1 if want_min else 0, dtype_code(dtype), self.stream()),
            "qamd_minmax",
        )
        return out


_DEFAULT = None


def default_device():
    """The process-wide device (``cuda:LOCAL_RANK`` under torchrun)."""
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = HipDevice()
    return _DEFAULT


def set_default_device(dev):
    """Install a device object (used by the test-suite to inject its numpy plan
    interpreter; the product never calls this with anything but HipDevice)."""
    global _DEFAULT
    _DEFAULT = dev
