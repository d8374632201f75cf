"""Execution options: ONE explicit, immutable object instead of process-global environment switches.

Round 4 selected kernels and executor behaviour through ~50 ``QAMD_*`` environment variables read at different moments
(plan time, launch time, in C and in Python) -- process-global, not thread-safe, and flipped mid-process by the benchmark
itself.  The contract of the boundary is "no hidden global state; reentrant".  Now:

* ``Options`` is a frozen dataclass.  A ``TreeExecutor`` / ``ContractExpression`` / ``ContractionProgram`` captures
  the options in force when it is BUILT and never looks anywhere else afterwards; per-call behaviour that a caller may
  want to vary (lanes, hipGraph replay of slices) is a keyword argument of the call.
* The default is a per-thread stack, as quimb keeps its contraction defaults (quimb/tensor/contraction.py:23-35):
  ``get_options()``, ``set_options(**kw)``, ``with options(**kw): ...``.
* The environment is consulted exactly ONCE, when this module is imported, to seed the process default
  (``Options.from_env``: a profiling script may still export ``QAMD_LANES=0`` before it starts Python).  Nothing reads
  ``os.environ`` after that; the C library takes its pins from the plan structs it is handed (include/quimb_amd.h:
  ``qamd_pair_plan.kernel`` / ``tile_cfg`` / ``split_k`` on input, ``QAMD_CHAIN2_FORCE_*`` flag bits).
"""

import contextlib
import dataclasses
import os
import threading


@dataclasses.dataclass(frozen=True)
class Options:
    # ---- plan building (TreeExecutor.__init__) ------------------------------------------------------------------------
    regroup: bool = True            #: (A.W1).W2 -> A.(W1.W2) where cheaper                          [QAMD_REGROUP]
    fuse_pairs: bool = True         #: two consecutive big-x-small steps -> one fused launch           [QAMD_CHAIN2]
    fuse_rows: bool = True          #: five site absorptions of a small boundary-sweep row -> one launch  [QAMD_ROWPASS]
    join_dot: bool = True           #: the closing inner product inside the last join's epilogue       [QAMD_JOIN_DOT]
    join_order: bool = True         #: issue the first join's chains first                             [QAMD_JOIN_ORDER]
    lane_priority: bool = False     #: stream priorities for the lanes (measured: a loss)              [QAMD_LANE_PRIORITY]
    hold_late: str = "auto"         #: hold later chains behind the first join's: "auto" | "0" | "1"   [QAMD_HOLD_LATE]
    program_join_order: bool = False  #: record launch programs in join order too                     [QAMD_PROGRAM_JOIN_ORDER]
    # ---- kernel pins (developer / test use; 0 / -1 / "auto" = the planner's choice) -----------------------------------
    row_kernel: str = "auto"        #: fused row: "auto" (= "quad": every row, the last one included) | "quad" (rowq.hip) | "tile" (rowpass.hip: small rows only)  [QAMD_ROW_KERNEL]
    chain2_kernel: str = "auto"     #: fused pair: "auto" | "lds" (chain2.hip) | "reg" (chain2r.hip) | "quad" (chain2q.hip, any size)
    pair_kernel: int = 0            #: qamd_pair_plan.kernel on input: 0 auto, -1 tiled GETT, -2 no MFMA GEMM kernels   [QAMD_KERNEL]
    tile_cfg: int = -1              #: qamd_pair_plan.tile_cfg on input                                 [QAMD_TILE_CFG]
    split_k: int = 0                #: qamd_pair_plan.split_k on input                                  [QAMD_SPLIT_K]
    join_arith: str = "f32"         #: arithmetic of the large fp32 GEMM-shaped joins: "f32" (fp32 MFMA, gemmk.hip) | "f16x3" (OPT-IN: the k-outer joins as split products on the f16 matrix pipe, fp32 accumulate, gemmh.hip: qamd_pair_plan.kernel = -7) | "f16x3-all" (OPT-IN: every large fp32 GEMM-shaped pair, any operand layout, complex pairs included: kernel = -8; on incoherent operands an fp32 chain's accuracy, not the blocked kernels')  [QAMD_JOIN_ARITH]
    # ---- execution ----------------------------------------------------------------------------------------------------
    lanes: bool = True              #: independent chains on their own HIP streams                      [QAMD_LANES]
    slice_graph: bool = True        #: slices replay one recorded hipGraph                              [QAMD_SLICE_GRAPH]
    lane_trace: bool = False        #: record an event per lane boundary (scripts/probes)               [QAMD_LANE_TRACE]
    program_own_lane0: bool = False  #: launch programs: lane 0 on a pool stream of its own            [QAMD_PROGRAM_OWN_LANE0]
    # ---- expressions --------------------------------------------------------------------------------------------------
    fold_constants: bool = True     #: contract constant-only sub-trees once                            [QAMD_FOLD_CONSTANTS]
    microtree: bool = True          #: trees of many small tensors walked by the device in one launch   [QAMD_MICROTREE]
    micro_wide: bool = True         #: fp32 / complex64 micro-trees carry their intermediates in fp64            [QAMD_MICRO_WIDE]
    micro_arena: str = "auto"       #: "auto" | "lds" | "global"                                        [QAMD_MICRO_ARENA]
    auto_program: bool = True       #: record a launch program on the 3rd call of an expression         [QAMD_AUTO_PROGRAM]
    auto_program_max_bytes: int = 4 << 30      #: [QAMD_AUTO_PROGRAM_MAX_BYTES]
    auto_program_total_bytes: int = 16 << 30   #: [QAMD_AUTO_PROGRAM_TOTAL_BYTES]
    debug: bool = False             #: say on stderr why a recording was refused                        [QAMD_DEBUG]

    def __post_init__(self):
        if self.join_arith not in ("f32", "f16x3", "f16x3-all"):
            raise ValueError(f"join_arith must be 'f32', 'f16x3' or 'f16x3-all', got {self.join_arith!r}")

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)

    @classmethod
    def from_env(cls, env=None):
        """The process default, seeded from ``QAMD_*`` variables ONCE (at import of this module)."""
        env = os.environ if env is None else env
        on = lambda k, d: (env.get(k, "1" if d else "0") != "0") if d else (env.get(k, "0") == "1")
        kw = dict(
            regroup=on("QAMD_REGROUP", True), fuse_pairs=on("QAMD_CHAIN2", True), fuse_rows=on("QAMD_ROWPASS", True), join_dot=on("QAMD_JOIN_DOT", True),
            join_order=on("QAMD_JOIN_ORDER", True), lane_priority=on("QAMD_LANE_PRIORITY", False),
            hold_late=env.get("QAMD_HOLD_LATE", "auto"), program_join_order=on("QAMD_PROGRAM_JOIN_ORDER", False),
            pair_kernel=int(env.get("QAMD_KERNEL", "0")), tile_cfg=int(env.get("QAMD_TILE_CFG", "-1")),
            split_k=int(env.get("QAMD_SPLIT_K", "0")), join_arith=env.get("QAMD_JOIN_ARITH", "f32"),
            lanes=on("QAMD_LANES", True), slice_graph=on("QAMD_SLICE_GRAPH", True), lane_trace=bool(env.get("QAMD_LANE_TRACE")),
            program_own_lane0=on("QAMD_PROGRAM_OWN_LANE0", False),
            fold_constants=on("QAMD_FOLD_CONSTANTS", True), microtree=on("QAMD_MICROTREE", True),
            micro_wide=on("QAMD_MICRO_WIDE", True), micro_arena=env.get("QAMD_MICRO_ARENA", "auto"), auto_program=on("QAMD_AUTO_PROGRAM", True),
            auto_program_max_bytes=int(env.get("QAMD_AUTO_PROGRAM_MAX_BYTES", str(4 << 30))),
            auto_program_total_bytes=int(env.get("QAMD_AUTO_PROGRAM_TOTAL_BYTES", str(16 << 30))),
            debug=bool(env.get("QAMD_DEBUG")), row_kernel=env.get("QAMD_ROW_KERNEL", "auto"),
        )
        c2 = "auto"
        if env.get("QAMD_CHAIN2R", "")[:1] == "0":
            c2 = "lds"
        elif env.get("QAMD_CHAIN2Q", "")[:1] == "0":
            c2 = "reg"
        elif env.get("QAMD_CHAIN2Q", "")[:1] == "2":
            c2 = "quad"
        return cls(chain2_kernel=c2, **kw)


_PROCESS_DEFAULT = Options.from_env()
_TLS = threading.local()


def _stack():
    st = getattr(_TLS, "stack", None)
    if st is None:
        st = _TLS.stack = [_PROCESS_DEFAULT]
    return st


def get_options():
    """The options new executors / expressions of THIS thread are built with."""
    return _stack()[-1]


def set_options(**kw):
    """Replace this thread's default; returns the previous ``Options``."""
    st = _stack()
    old = st[-1]
    st[-1] = old.replace(**kw)
    return old


@contextlib.contextmanager
def options(**kw):
    """``with quimb_amd.options(lanes=False): ...`` -- a scoped default for this thread."""
    st = _stack()
    st.append(st[-1].replace(**kw))
    try:
        yield st[-1]
    finally:
        st.pop()
