"""A HipDevice that can RECORD launch programs without a GPU (CPU tests of quimb_amd/program.py's plumbing).

Recording never launches anything -- the library appends every call to the program -- so the only parts of ``HipDevice``
that need a device are its constructor's GPU check, plan-table builds (a real kernel) and synchronisation.  This subclass
stubs exactly those: "device memory" is host torch tensors (their addresses are only recorded, never dereferenced), the
k-offset tables stay unfilled.  Running a program is not possible here; the ``-m gpu`` tests do that."""
import ctypes as C
import os

import numpy as np

from quimb_amd import _lib
from quimb_amd.device import HipDevice, _CompiledPair, dtype_code, fill_plan_struct


class RecordOnlyDevice(HipDevice):
    def __init__(self):
        import torch

        self.lib = _lib.load()
        self.torch = torch
        self.index = 0
        self.tdev = torch.device("cpu")
        self._tdt = {np.dtype("float32"): torch.float32, np.dtype("float64"): torch.float64,
                     np.dtype("complex64"): torch.complex64, np.dtype("complex128"): torch.complex128,
                     np.dtype("int64"): torch.int64}
        self._pairs = {}
        self.record = None
        self.profile = None
        self.profile_min_mults = 0
        self.force_tile_cfg, self.force_split_k, self.force_kernel = -1, 0, 0
        self.force_chain2, self.micro_arena = "auto", "auto"

    def stream(self):
        return None

    def synchronize(self):
        pass

    def lane_streams(self, n):
        raise RuntimeError("no streams on the record-only device")

    def compile_pair(self, spec, dtype, align_a=16, align_b=16, align_c=16):
        code = dtype_code(dtype)
        key = (spec, code, align_a, align_b, align_c)
        cp = self._pairs.get(key)
        if cp is None:
            p = fill_plan_struct(spec, code)
            p.tile_cfg, p.split_k, p.kernel = self.force_tile_cfg, self.force_split_k, self.force_kernel
            _lib.check(self.lib.qamd_pair_plan_finalize(C.byref(p), align_a, align_b, align_c), "qamd_pair_plan_finalize")
            cp = _CompiledPair()
            cp.struct = p
            cp.ktab = self.torch.zeros(int(self.lib.qamd_pair_ktab_len(C.byref(p))), dtype=self.torch.int64)   # not built
            cp.ws_bytes = int(self.lib.qamd_pair_workspace_bytes(C.byref(p)))
            cp.ready = cp.ready_stream = None
            self._pairs[key] = cp
        return cp
