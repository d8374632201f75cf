"""Host-only: the plan of the busiest rank's share of a WORLD-rank job, step by step -- lane, kind, dims, the kernel the
planner picks, algorithmic bytes and the HBM-floor time at 4.5 TB/s.
    python scripts/probes/share_plan.py [WORLD]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import quimb_amd as qa
from bench import build_network
from quimb_amd import _lib
from quimb_amd.device import dtype_code, fill_plan_struct
from quimb_amd.quadrants import QuadrantRank, QuadrantSharding

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
arrays, inputs, size = build_network(10, 10, 6, 7, "float32")
if world == 1:
    ex = qa.TreeExecutor(qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(10, 10)), "float32")
else:
    sh = QuadrantSharding(inputs, size, 10, 10, world)
    r = int(np.argmax(sh.cost_report()["per_rank_mults"]))
    ex = QuadrantRank(sh, r, "float32").executor
lib = _lib.load()


def describe(spec):
    p = fill_plan_struct(spec, dtype_code(np.dtype("float32")))
    p.tile_cfg, p.split_k, p.kernel = -1, 0, 0
    if lib.qamd_pair_plan_finalize(C.byref(p), 16, 16, 16):
        return "?"
    buf = C.create_string_buffer(200)
    lib.qamd_pair_describe(C.byref(p), buf, 200)
    return buf.value.decode()


def describe_chain2(c2):
    """the fused-pair kernel the library would launch (16-byte aligned result assumed)"""
    pl = _lib.Chain2PlanStruct()
    pl.dtype, pl.D, pl.nm = 0, c2.D, len(c2.m)
    for i, (d, sa, sc) in enumerate(c2.m):
        pl.dim_m[i], pl.sa_m[i], pl.sc_m[i] = d, sa, sc
    pl.sa_v = c2.sa_v
    if all(o % 4 == 0 for o in c2.off_co) and all(sc % 4 == 0 for (_, _, sc) in c2.m[:-1]):
        pl.flags = 1
    if c2.k1_single:
        pl.flags |= 2
    if c2.no_n2out:
        pl.flags |= 4
    buf = C.create_string_buffer(128)
    lib.qamd_chain2_describe(C.byref(pl), buf, 128)
    return buf.value.decode()


tot = {}
for i, (e, info) in enumerate(zip(ex.plan, ex.info)):
    lane = ex.lanes[i] if hasattr(ex, "lanes") else 0
    if e[0] == "pair":
        st = e[4]
        name = describe(st.spec) if st.kind == "gett" else st.kind
        g = st.spec if st.kind == "gett" else None
        dims = (g.B, g.M, g.N, g.K) if g is not None else (info.B, info.M, info.N, info.K)
        pre = "pre" if st.kind == "gett" and any(st.pre) else ""
    elif e[0] == "chain2":
        c2 = e[5]
        name, dims, pre = f"{describe_chain2(c2)} M={c2.M}", (info.B, info.M, info.N, info.K), ""
    else:
        name, dims, pre = e[0], (info.B, info.M, info.N, info.K), ""
    us = info.bytes / 4.5e6
    tot[lane] = tot.get(lane, 0) + us
    print(f"{i:3d} lane {lane} {e[0]:7s} mults {info.mults:12d} bytes {info.bytes:11d} floor {us:7.1f} us  dims {dims}  {name} {pre}")
print("floor per lane (us):", {k: round(v, 1) for k, v in tot.items()})
