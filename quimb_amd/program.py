"""Launch programs: one contraction's launch sequence recorded once, replayed with ONE host call.

quimb caches the planned contraction expression and re-runs cotengra's per-step Python loop on every call
(quimb/tensor/contraction.py:285, cache pinned by tests/test_tensor/test_contract.py:155-172).  The first rounds of
this backend kept that loop in Python: ~15 us of interpreter + ctypes time per launch, 1.6 ms for the ~100 launches
of one rank's share of the headline network -- more than the device needs for that share's corner sweeps, so the
host, not the GPU, set the pace (profiles/r03_rank_of_8_timeline_profiled.txt).  A ``ContractionProgram`` runs the
executor ONCE with the library in recording mode (csrc/program.cpp: every ``qamd_*`` launch is appended to a C-side
list instead of issued; lanes and cross-lane waits are recorded as such), keeps every intermediate buffer of that run
alive at its address, and from then on ``program(arrays)`` is one ``qamd_program_run`` call: the recorded launches on
their lanes' HIP streams, input pointers re-based onto the caller's arrays, no Python per step, no allocation.

Memory: the recorder owns a pool of device blocks.  A block is reused only by a LATER launch of the SAME lane (stream
order makes that safe at replay); buffers handed from one lane to another are never reused.  Inputs are not copied.
"""

import ctypes as C
import os

import numpy as np

from . import _lib
from .array import Array, asarray


class RecordPool:
    """Device blocks of one recording: ``alloc`` hands out a typed view of a block, ``release`` returns the block to
    the free list of the lane that is current at that moment (the lane of the launch that used it last)."""

    def __init__(self, torch, tdev):
        self.torch, self.tdev = torch, tdev
        self.blocks = {}        # base address -> uint8 block tensor (every block ever made: alive with the program)
        self.free = {}          # lane -> [(nbytes, base address)]
        self.in_use = set()
        self.lane = 0
        self.bytes_total = 0

    def alloc(self, n, tdtype):
        isz = self.torch.empty(0, dtype=tdtype).element_size()
        want = max(int(n) * isz, 1)
        fl = self.free.setdefault(self.lane, [])
        best = None
        for k, (nb, base) in enumerate(fl):          # smallest block that fits and is not wastefully large
            if want <= nb <= max(4 * want, want + (1 << 16)) and (best is None or nb < fl[best][0]):
                best = k
        if best is not None:
            nb, base = fl.pop(best)
            block = self.blocks[base]
        else:
            nb = -(-want // 512) * 512
            block = self.torch.empty(nb, dtype=self.torch.uint8, device=self.tdev)
            base = block.data_ptr()
            self.blocks[base] = block
            self.bytes_total += nb
        self.in_use.add(base)
        return block[: int(n) * isz].view(tdtype) if n else block[:isz].view(tdtype)

    def release(self, t):
        """``t``: a tensor ``alloc`` returned (or any view of it).  Unknown tensors (inputs, plan tables) are ignored."""
        base = t.untyped_storage().data_ptr()
        if base in self.in_use:
            self.in_use.discard(base)
            self.free.setdefault(self.lane, []).append((self.blocks[base].numel(), base))


class _Recorder:
    """What ``HipDevice.record`` points at while a program is being recorded."""

    def __init__(self, dev, prog, nlanes, mark_min_mults):
        self.dev, self.prog, self.nlanes = dev, prog, nlanes
        self.pool = RecordPool(dev.torch, dev.tdev)
        self.mark_min_mults = mark_min_mults
        self.marked = []        # tag -> (spec, dtype, kernel name, split_k)

    # memory
    def alloc(self, n, tdtype):
        return self.pool.alloc(n, tdtype)

    def release(self, t):
        self.pool.release(t)

    # lanes
    def set_lane(self, lane):
        self.pool.lane = lane
        _lib.check(self.dev.lib.qamd_program_set_lane(self.prog, lane), "qamd_program_set_lane")

    def wait(self, lane, on_lane):
        _lib.check(self.dev.lib.qamd_program_wait(self.prog, lane, on_lane), "qamd_program_wait")

    # timing marks: the launches that can be a step's dominant kernel
    def maybe_mark(self, spec, dtype, describe):
        if self.mark_min_mults is None or getattr(spec, "mults", 0) < self.mark_min_mults:
            return
        name, split_k = describe()
        _lib.check(self.dev.lib.qamd_program_mark(self.prog, len(self.marked)), "qamd_program_mark")
        self.marked.append((spec, dtype, name, split_k))


class _Elapsed:
    """Stands in for a HIP event pair in ``HipDevice.profile``-style records: ``a.elapsed_time(b)`` -> b's reading."""

    def __init__(self, ms=0.0):
        self.ms = ms

    def elapsed_time(self, other):
        return other.ms


class ContractionProgram:
    """``TreeExecutor.program(arrays, strip_exponent)``: record now, replay with ``program()`` / ``program(arrays)``.

    The result buffers (and the exponent accumulator) belong to the program and are overwritten by the next run --
    ``.copy()`` what must outlive it.  Unsliced trees on the HIP device only."""

    def __init__(self, executor, arrays, strip_exponent=False, mark_min_mults=None):
        if executor.tree.nslices != 1:
            raise ValueError("launch programs record unsliced trees")
        self.executor, self.strip_exponent = executor, bool(strip_exponent)
        xs = list(executor.check_inputs(arrays))
        dev = xs[0]._dev
        if not hasattr(dev, "lib") or not hasattr(dev, "torch"):
            raise RuntimeError("launch programs need the HIP device")
        # Replay re-bases every recorded operand pointer onto "the input whose address range contained it".  That is only
        # well defined when the ranges of the recording's inputs are DISJOINT: record ``expr(A, A)`` (or <psi|psi> with
        # ``conj()`` returning self for real data, or overlapping views) and every read of the second input would be
        # re-based onto the first at replay -- ``expr(A, B)`` would silently return the value for (A, A).  An input that
        # overlaps an earlier one is therefore recorded on a private copy (replays may alias freely: inputs are only read).
        spans = []
        for i, x in enumerate(xs):
            lo = x._buf.data_ptr()
            hi = lo + max(x.size, 1) * x.dtype.itemsize
            if any(lo < h and l < hi for l, h in spans):
                x = xs[i] = x.copy()
                lo = x._buf.data_ptr()
                hi = lo + max(x.size, 1) * x.dtype.itemsize
            spans.append((lo, hi))
        if dev.record is not None:
            raise RuntimeError("a launch program is already being recorded on this device")
        self._dev = dev
        self.inputs = xs                                    # the arrays the recording ran on (the default inputs)
        self._shapes = [x.shape for x in xs]
        self.nlanes = max(int(getattr(executor, "nlanes", 1)), 1)
        self._priorities = list(getattr(executor, "lane_priority", [0] * self.nlanes))
        lib = dev.lib
        self._prog = lib.qamd_program_create(self.nlanes)
        if not self._prog:
            raise _lib.QamdError("qamd_program_create failed")
        rec = _Recorder(dev, self._prog, self.nlanes, mark_min_mults)
        _lib.check(lib.qamd_program_record_begin(self._prog), "qamd_program_record_begin")
        dev.record = rec
        try:
            self._exponent = dev.new_exponent() if strip_exponent else None
            self.output = executor._run_core(xs, self._exponent, None, lanes=True)
        finally:
            dev.record = None
            _lib.check(lib.qamd_program_record_end(self._prog), "qamd_program_record_end")
        self._pool = rec.pool                               # owns every intermediate for the program's lifetime
        self.marked = rec.marked
        n = len(xs)
        self._in_ptrs0 = [x._buf.data_ptr() for x in xs]
        ptrs = (C.c_void_p * max(n, 1))(*self._in_ptrs0)
        nbytes = (C.c_int64 * max(n, 1))(*[max(x.size, 1) * x.dtype.itemsize for x in xs])
        _lib.check(lib.qamd_program_bind_inputs(self._prog, n, ptrs, nbytes), "qamd_program_bind_inputs")
        self._ptrs0 = ptrs
        self._keep = self._keep_src = self._keep_ptrs = None
        self.num_launches = int(lib.qamd_program_num_launches(self._prog))
        self.num_ops = int(lib.qamd_program_num_ops(self._prog))
        dev.synchronize()        # plan tables compiled during the recording are final before the first replay

    def __del__(self):
        prog, self._prog = getattr(self, "_prog", None), None
        if prog:
            try:
                self._dev.synchronize()      # side lanes may still be reading the pool's blocks
            except Exception:
                pass
            self._dev.lib.qamd_program_destroy(prog)

    def forget_inputs(self):
        """Drop the references to the arrays recorded on (a program that is always called WITH arrays need not keep the
        first call's alive); ``program()`` without arrays is not possible afterwards."""
        self.inputs = [None] * len(self.inputs)
        self._ptrs0 = None

    @property
    def pool_bytes(self):
        return self._pool.bytes_total

    def __call__(self, arrays=None, defer_exponent=False, timing_slot=None):
        """Run.  ``arrays``: device arrays of the recorded shapes / dtype (default: the arrays recorded on); they are
        read in place.  Returns the output ``Array`` (or ``(Array, exponent)``); ``defer_exponent`` hands the
        device-resident accumulator back instead of reading it (no synchronisation).  ``timing_slot``: record the
        marked launches' durations into that slot (``timings(slot)``)."""
        dev = self._dev
        if timing_slot is None:
            timing_slot = getattr(self, "_timing_slot", None)
        if arrays is None:
            if self._ptrs0 is None:
                raise ValueError("this program has forgotten the arrays it was recorded on: pass arrays")
            ptrs = self._ptrs0
        elif (self._keep_src is not None and len(arrays) == len(self._keep_src)
              and all(a is b for a, b in zip(arrays, self._keep_src))):
            ptrs = self._keep_ptrs                            # the same DEVICE array objects as last time: nothing to check again
        else:
            if len(arrays) != len(self.inputs):
                raise ValueError(f"expected {len(self.inputs)} arrays, got {len(arrays)}")
            xs = []
            for x, shape, p0 in zip(arrays, self._shapes, self._in_ptrs0):
                x = asarray(x).astype(self.executor.dtype)
                if x.shape != shape:
                    raise ValueError(f"array shape {x.shape} does not match the recorded {shape}")
                if (x._buf.data_ptr() - p0) % 16:
                    x = x.copy()       # the recorded vector widths assume the recorded alignment class
                    if (x._buf.data_ptr() - p0) % 16:
                        raise _lib.QamdError("input alignment differs from the recorded program's")
                xs.append(x)
            ptrs = (C.c_void_p * max(len(xs), 1))(*[x._buf.data_ptr() for x in xs])
            self._keep = xs                                  # until the next run: the launches are asynchronous
            # the identity shortcut above is for device arrays only: a host (numpy) input was uploaded just now, and the
            # same ndarray object may have been written to by the next call -- it is uploaded again every time
            device_in = all(isinstance(a, Array) and a is x for a, x in zip(arrays, xs))
            self._keep_src, self._keep_ptrs = (list(arrays), ptrs) if device_in else (None, None)
        # lane 0 = the caller's stream (as in launch-by-launch execution); options.program_own_lane0: a pool stream of its
        # own, forked from / joined to the caller's (what lane priorities need)
        own0 = bool(getattr(self.executor, "options", None) and self.executor.options.program_own_lane0)
        streams = dev.lane_streams(self.nlanes, self._priorities, own_lane0=own0)
        caller = dev.torch.cuda.current_stream(dev.tdev)
        # consecutive replays share the pool and the output buffer: stream order protects them on ONE caller stream; a
        # caller that moved to another stream first waits for the previous replay's lane 0
        last = getattr(self, "_last_caller", None)
        if last is not None and last != caller:
            caller.wait_stream(last)
        self._last_caller = caller
        if own0:
            streams[0].wait_stream(caller)
        sarr = (C.c_void_p * self.nlanes)(*[s.cuda_stream for s in streams])
        _lib.check(dev.lib.qamd_program_run(self._prog, sarr, ptrs, 0 if timing_slot is None else int(timing_slot) + 1),
                   "qamd_program_run")
        if own0:
            caller.wait_stream(streams[0])
        if self.strip_exponent:
            return self.output, (self._exponent if defer_exponent else dev.read_exponent(self._exponent))
        return self.output

    def timings(self, slot=0):
        """[(spec, dtype, kernel name, split_k, start, end)] of the marked launches of the last run that used
        ``slot`` -- the shape of ``HipDevice.profile`` records (``start.elapsed_time(end)`` = milliseconds).
        Synchronises."""
        self._dev.synchronize()
        out = []
        tag, ms = C.c_int32(), C.c_float()
        for i in range(int(self._dev.lib.qamd_program_num_marks(self._prog))):
            rc = self._dev.lib.qamd_program_mark_ms(self._prog, i, int(slot), C.byref(tag), C.byref(ms))
            if rc:
                continue
            spec, dtype, name, sk = self.marked[tag.value]
            out.append((spec, dtype, name, sk, _Elapsed(), _Elapsed(float(ms.value))))
        return out


class EagerProgram:
    """The ``ContractionProgram`` interface on a device without the launch recorder (the test-suite's plan interpreter):
    every call re-executes the plan.  Host logic written against programs -- the bench's timing slots, a rank's step
    with its collective -- then runs unchanged in the CPU tests; the product's device always records."""

    def __init__(self, executor, arrays, strip_exponent=False, mark_min_mults=None):
        if executor.tree.nslices != 1:
            raise ValueError("launch programs record unsliced trees")
        self.executor, self.strip_exponent = executor, bool(strip_exponent)
        self.inputs = executor.check_inputs(arrays)
        self.nlanes = max(int(getattr(executor, "nlanes", 1)), 1)
        self.marked, self.num_launches, self.num_ops, self.pool_bytes = [], len(executor.plan), len(executor.plan), 0

    def __call__(self, arrays=None, defer_exponent=False, timing_slot=None):
        xs = self.inputs if arrays is None else arrays
        if self.strip_exponent:
            return self.executor(xs, strip_exponent=True, defer_exponent=defer_exponent)
        return self.executor(xs)

    def timings(self, slot=0):
        return []
