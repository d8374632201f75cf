"""Whole-tree executor: device-resident evaluation of a contraction tree.

This is the MI355X replacement for the per-step python loop inside
``cotengra.array_contract`` that quimb calls at quimb/tensor/contraction.py:285
(hot loop described in SURVEY.md section 3.1).  Differences by design:

* the executor owns every intermediate, so it *chooses* each intermediate's
  index order such that the next contraction needs no permute (the result takes
  the big operand's layout with the contracted run replaced by the small
  operand's free indices) -- the reference materialises a transpose per step;
* the final step writes straight into the requested output order;
* ``strip_exponent`` (quimb/tensor/tensor_core.py:330-340) runs entirely on the
  device: per-step max|x| and the log10 accumulator never visit the host;
* sliced indices (cotengra ``tree.sliced_inds``) are executed as independent
  slices whose outputs sum to the result; sub-trees that do not depend on any
  sliced index are evaluated once and reused by every slice.
"""

import os

import contextlib

import numpy as np

from .array import Array, asarray, _coerce_dtype
from .ops import _einsum_single, run_pair_step
from .options import get_options
from .pairwise import ROWPASS_MAX_OUT, ROWPASS_SITES, plan_chain2, plan_pair, plan_rowpass, prod
from .tree import ContractionTree


#: a first join of at least this many multiplications is long enough for the later joins' chains to run beside it
#: (``TreeExecutor._order_for_joins``): between the 2.35e11 of a rank of two (loses) and the 4.7e11 of the whole network (gains)
HOLD_LATE_MIN_MULTS = 3.5e11


class StepInfo:
    __slots__ = ("kind", "mults", "bytes", "M", "N", "K", "B", "sliced_dep")

    def __init__(self, kind, mults, nbytes, dims, dep):
        self.kind, self.mults, self.bytes = kind, mults, nbytes
        self.B, self.M, self.N, self.K = dims
        self.sliced_dep = dep


class TreeExecutor:
    """Plan once, run many times (the analogue of a cached cotengra expression,
    pinned by tests/test_tensor/test_contract.py:155-172 in the reference)."""

    def __init__(self, tree: ContractionTree, dtype="float32", join_order=True, options=None):
        #: issue the chains of the FIRST join first and that join right behind them (``_order_for_joins``): right for the
        #: launch-by-launch path, where the host enqueues slower than the device executes; a launch program enqueues a
        #: whole share in 0.35 ms and is recorded in plain plan order (all chains side by side, then the joins)
        #: the options this executor was BUILT with (quimb_amd/options.py): resolved once, here; nothing below or at run
        #: time consults the environment
        self.options = options if options is not None else get_options()
        self.join_order = bool(join_order)
        if self.options.regroup:
            tree = tree.regrouped()          # (A.W1).W2 -> A.(W1.W2) where cheaper (sliced bonds); same result
        self.tree = tree
        self.dtype = _coerce_dtype(dtype)
        size = tree.size_dict
        sliced = set(tree.sliced_inds)
        n = len(tree.inputs)
        self.input_inds = [tuple(ix for ix in t if ix not in sliced) for t in tree.inputs]
        self.input_sliced_axes = [
            tuple((ax, ix) for ax, ix in enumerate(t) if ix in sliced) for t in tree.inputs
        ]
        layout = {i: self.input_inds[i] for i in range(n)}
        dep = {i: bool(self.input_sliced_axes[i]) for i in range(n)}
        death = tree.death_times()
        self.plan = []
        self.info = []
        nsteps = len(tree.steps)
        isz = self.dtype.itemsize
        for si, (con, res, ops, keep, _) in enumerate(tree.steps):
            last = si == nsteps - 1 and len(tree.remaining) == 1
            if len(con) == 1:
                src = layout[con[0]]
                out = tuple(tree.output) if last else tuple(keep)
                self.plan.append(("single", con[0], res, src, out))
                layout[res] = out
                dep[res] = dep[con[0]]
                m = prod(size[ix] for ix in set(src))
                self.info.append(StepInfo("single", 0, isz * (m + prod(size[ix] for ix in out)), (1, 1, 1, 1), dep[res]))
                continue
            if len(con) != 2:
                raise ValueError("only pairwise (or single-operand) contraction steps are supported")
            a, b = con
            la, lb = layout[a], layout[b]
            sa = tuple(size[ix] for ix in la)
            sb = tuple(size[ix] for ix in lb)
            if last:
                try:
                    step = plan_pair(la, sa, lb, sb, tuple(tree.output), True, None)
                except NotImplementedError:
                    # the caller's output order interleaves more index groups than one launch addresses:
                    # contract into the kernel's own order, then one permute pass into the requested one
                    step = plan_pair(la, sa, lb, sb, tuple(tree.output), False, None)
                    tmp = ("unordered", res)
                    self.plan.append(("pair", a, b, tmp, step))
                    g = step.spec
                    self.info.append(StepInfo(step.kind, step.mults,
                                              isz * g.B * (g.M * g.K + g.K * g.N + g.M * g.N), (g.B, g.M, g.N, g.K),
                                              dep[a] or dep[b]))
                    out = tuple(tree.output)
                    self.plan.append(("single", tmp, res, step.out_inds, out))
                    layout[res] = out
                    dep[res] = dep[tmp] = dep[a] or dep[b]
                    self.info.append(StepInfo("single", 0, 2 * isz * prod(step.out_shape), (1, 1, 1, 1), dep[res]))
                    continue
            else:
                d = tuple(sorted(death.get(res, {}).items(), key=repr))
                step = plan_pair(la, sa, lb, sb, tuple(keep), False, d)
            self.plan.append(("pair", a, b, res, step))
            layout[res] = step.out_inds
            dep[res] = dep[a] or dep[b]
            if step.kind == "gett":
                g = step.spec
                dims = (g.B, g.M, g.N, g.K)
                nbytes = isz * g.B * (g.M * g.K + g.K * g.N + g.M * g.N)
            else:
                dims = (prod(step.out_shape), 1, 1, 1)
                nbytes = isz * 3 * prod(step.out_shape)
            self.info.append(StepInfo(step.kind, step.mults, nbytes, dims, dep[res]))
        self.layout = layout
        self.dep = dep
        self._fuse_rows(size)
        self._fuse_pairs(size)
        self._fuse_join_dot()
        self._assign_lanes()
        if len(tree.remaining) != 1:
            raise ValueError("contraction path does not reduce the network to a single tensor")
        (self.root,) = tree.remaining
        self.out_inds = layout[self.root]
        if n == 1 and not tree.steps:
            self.out_inds = tuple(tree.output)
        self._hoisted = None  # cache of slice-independent intermediates for the current inputs

    def _fuse_rows(self, size):
        """Five consecutive site absorptions of a SMALL boundary-sweep row (each step's result the next one's big operand)
        become one launch (``qamd_contract_rowpass``, csrc/rowpass.hip): in the first rows of a corner sweep every step
        is ~12 us of dispatch latency around a few microseconds of work, and five of them depend on each other.  The fused
        entry reads and writes the layouts the five steps had; rows too large to be latency-bound (``ROWPASS_MAX_OUT``)
        and everything that does not have the row structure (``plan_rowpass``) stay as they were -- fused pairs take
        the large rows.  ``options.fuse_rows = False`` keeps every step a separate launch."""
        if self.dtype != np.dtype("float32") or not self.options.fuse_rows:
            return
        plan, info = self.plan, self.info
        rk = self.options.row_kernel     # "auto" / "quad": rowq.hip, rows of any size; "tile": rowpass.hip, small rows only
        uses = {}
        for e in plan:
            for o in self._entry_io(e)[0]:
                uses[o] = uses.get(o, 0) + 1
        isz = self.dtype.itemsize

        def big_small(entry):
            if entry[0] != "pair":
                return None
            _, a, b, r, st = entry
            if st.kind != "gett" or any(st.pre) or st.spec.b:
                return None
            return ((b, a) if st.swapped else (a, b)) + (r, st)

        # pass 1: which runs of five steps have the row structure (layouts as planned step by step)
        runs = {}
        i = 0
        while i < len(plan):
            run = []
            for j in range(i, min(i + ROWPASS_SITES, len(plan))):
                bs = big_small(plan[j])
                if bs is None or (run and (bs[0] != run[-1][2] or uses.get(bs[0], 0) != 1
                                           or self.dep[bs[2]] != self.dep[run[0][2]])):
                    break
                run.append(bs)
            rp = None
            if len(run) == ROWPASS_SITES:
                rp = plan_rowpass(self.layout[run[0][0]], [self.layout[r[1]] for r in run], run[-1][3].out_inds, size,
                                  self.dtype.name, rk)
            if rp is not None and rp.c_size > ROWPASS_MAX_OUT and i + ROWPASS_SITES < len(plan):
                # a LARGE row (bandwidth, not latency) is only worth one launch when it is the whole row: the entry
                # cannot start mid-row, so a wider row would be split 5 + rest and the rest would lose its fused pairs
                nxt = big_small(plan[i + ROWPASS_SITES])
                if nxt is not None and nxt[0] == run[-1][2] and set(self.layout[nxt[1]]) & set(self.layout[run[-1][1]]):
                    rp = None
            if rp is not None:
                runs[i] = run
                i += ROWPASS_SITES
            else:
                i += 1
        # the FIRST row of a sweep: W0 . W1 (two site tensors), then three more absorptions -- four launches for a product of
        # five small tensors along their bonds; one launch of the same entry without a boundary tensor
        firsts = {}
        i = 0
        while i + ROWPASS_SITES - 1 <= len(plan):
            if i in runs or any(k in runs for k in range(max(i - ROWPASS_SITES + 1, 0), i)):
                i += 1
                continue
            chain = []          # (big or previous result, site, result, step): roles by identity, not by size -- the
            for j in range(i, min(i + ROWPASS_SITES - 1, len(plan))):      # first row's tensors are all about as small
                bs = big_small(plan[j])
                if bs is None:
                    break
                if chain:
                    prev = chain[-1][2]
                    if prev not in bs[:2] or uses.get(prev, 0) != 1 or self.dep[bs[2]] != self.dep[chain[0][2]]:
                        break
                    bs = (prev, bs[1] if bs[0] == prev else bs[0]) + bs[2:]
                chain.append(bs)
            if len(chain) == ROWPASS_SITES - 1:
                a0, b0 = chain[0][0], chain[0][1]
                if len(self.layout[a0]) > len(self.layout[b0]):
                    a0, b0 = b0, a0                                   # the corner site (two legs) starts the row
                sites = [a0, b0] + [c_[1] for c_ in chain[1:]]
                if plan_rowpass(None, [self.layout[w_] for w_ in sites], chain[-1][3].out_inds, size,
                                self.dtype.name) is not None:
                    firsts[i] = (chain, sites)
                    i += ROWPASS_SITES - 1
                    continue
            i += 1
        # pass 2: a fused row whose result feeds the NEXT fused row (and nothing else) is free to write it in the order
        # that serves both kernels -- spectators and the new open leg outermost, the five new down legs innermost: the
        # producer then stores 5 KB runs and the consumer's work items read 31 KB contiguous each, instead of 24-byte runs
        # and a 7776-line gather in the death-ordered layout.  (The last fused row keeps the planned layout: what follows
        # it -- fused pairs, streaming kernels -- was planned against that.)
        by_big = {run[0][0]: i_ for i_, run in runs.items()}
        new_plan, new_info = [], []
        relaid = set()          # results whose layout pass 2 changed from the step-by-step plan's
        i = 0
        def next_row_layout(res, lc):
            """``lc`` reordered for a fused consumer: everything else outermost, the consumer's five up legs innermost"""
            if res in by_big and uses.get(res, 0) == 1:
                nxt_sites = set()
                for r in runs[by_big[res]]:
                    nxt_sites |= set(self.layout[r[1]])
                downs = [ix for ix in lc if ix in nxt_sites]
                if len(downs) == ROWPASS_SITES:
                    new = tuple(ix for ix in lc if ix not in nxt_sites) + tuple(downs)
                    # commit the new order only if the CONSUMER row still plans against it (with the layout its own
                    # result will get): its separate steps were planned against the old order and could not be kept
                    crun = runs[by_big[res]]
                    clc = next_row_layout(crun[-1][2], tuple(crun[-1][3].out_inds))
                    if plan_rowpass(new, [self.layout[r[1]] for r in crun], clc, size, self.dtype.name, rk) is not None:
                        return new
            return lc

        while i < len(plan):
            if i in firsts:
                chain, sites = firsts[i]
                res = chain[-1][2]
                lc0 = tuple(chain[-1][3].out_inds)
                rp = plan_rowpass(None, [self.layout[w_] for w_ in sites], next_row_layout(res, lc0), size, self.dtype.name)
                if rp is None:
                    rp = plan_rowpass(None, [self.layout[w_] for w_ in sites], lc0, size, self.dtype.name)
                if rp is not None:
                    if rp.out_inds != lc0:
                        relaid.add(res)
                    self.layout[res] = rp.out_inds
                    new_plan.append(("rowpass", None, tuple(sites), res, rp))
                    new_info.append(StepInfo("rowpass", rp.mults, isz * (rp.c_size + ROWPASS_SITES * rp.D**3),
                                             (1, rp.c_size // rp.D**2, rp.D**2, rp.D), self.dep[res]))
                    i += ROWPASS_SITES - 1
                    continue
            run = runs.get(i)
            rp = None
            if run is not None:
                res = run[-1][2]
                lc0 = tuple(run[-1][3].out_inds)
                rp = plan_rowpass(self.layout[run[0][0]], [self.layout[r[1]] for r in run], next_row_layout(res, lc0), size,
                                  self.dtype.name, rk)
                if rp is None:
                    rp = plan_rowpass(self.layout[run[0][0]], [self.layout[r[1]] for r in run], lc0, size, self.dtype.name, rk)
                if rp is None and run[0][0] in relaid:
                    # (cannot happen: next_row_layout re-planned this row against the new order before its producer
                    # committed it) -- the separate steps were planned against the OLD order of the operand
                    raise AssertionError("fused row: the consumer of a re-ordered row result no longer plans")
            if rp is not None:
                res = run[-1][2]
                if rp.out_inds != lc0:
                    relaid.add(res)
                self.layout[res] = rp.out_inds
                new_plan.append(("rowpass", run[0][0], tuple(r[1] for r in run), res, rp))
                new_info.append(StepInfo("rowpass", rp.mults, isz * (rp.a_size + rp.c_size + ROWPASS_SITES * rp.D**4),
                                         (1, rp.c_size // rp.D**2, rp.D**2, rp.D**2), self.dep[res]))
                i += ROWPASS_SITES
            else:
                new_plan.append(plan[i])
                new_info.append(info[i])
                i += 1
        self.plan, self.info = new_plan, new_info

    def _fuse_pairs(self, size):
        """Replace consecutive big-x-small steps that have the two-site structure by one
        fused launch (qamd_contract_chain2): the intermediate never reaches HBM."""
        # on by default: 0.94 ms per fused 6^9 pair against 2 x 0.55 ms for two streaming
        # launches; options.fuse_pairs = False keeps every step a separate launch
        if self.dtype.kind == "c" or not self.options.fuse_pairs:
            return
        plan, info = self.plan, self.info
        new_plan, new_info = [], []
        isz = self.dtype.itemsize
        def big_small(entry):
            """(big operand id, small operand id, result id, step) of a plain big-x-small GETT step, else None"""
            if entry[0] != "pair":
                return None
            _, a, b, r, st = entry
            if st.kind != "gett" or any(st.pre) or st.spec.b:
                return None
            return ((b, a) if st.swapped else (a, b)) + (r, st)

        i = 0
        while i < len(plan):
            fused = None
            s1 = big_small(plan[i])
            s2 = big_small(plan[i + 1]) if i + 1 < len(plan) else None
            if s1 and s2 and s2[0] == s1[2] and self.dep[s1[2]] == self.dep[s2[2]]:
                c2 = plan_chain2(self.layout[s1[0]], self.layout[s1[1]], s1[3].out_inds, self.layout[s2[1]],
                                 s2[3].out_inds, size, self.dtype.name, variants=self.options.chain2_kernel != "lds")
                if c2 is not None:
                    fused = ("chain2", s1[0], s1[1], s2[1], s2[2], c2)
            if fused is not None:
                c2 = fused[5]
                new_plan.append(fused)
                new_info.append(StepInfo("chain2", c2.mults, isz * (c2.a_size + c2.c_size + 2 * c2.D**4),
                                         (1, c2.M, c2.D**2, c2.D**2), self.dep[fused[4]]))
                i += 2
            else:
                new_plan.append(plan[i])
                new_info.append(info[i])
                i += 1
        self.plan, self.info = new_plan, new_info

    def _fuse_join_dot(self):
        """The closing two steps of a two-sided contraction -- a GEMM-shaped join whose result then meets a tensor of the
        SAME layout in one inner product over all indices (the quadrant tree: B = BL.BR, then sum(T * B)) -- become one
        plan entry: the device multiplies every result tile with the matching tile of ``t`` and sums, so the join's result
        is never written or read back (``qamd_contract_pair_dot``; fp32 joins on the k-outer MFMA kernel, anything else is
        executed as the two steps it was).  ``options.join_dot = False`` keeps the steps apart."""
        if self.dtype != np.dtype("float32") or self.tree.nslices != 1 or len(self.plan) < 2 \
                or not self.options.join_dot:
            return
        last, prev = self.plan[-1], self.plan[-2]
        if last[0] != "pair" or prev[0] != "pair":
            return
        _, da, db, dres, dstep = last
        _, ja, jb, jres, jstep = prev
        if dstep.kind != "gett" or jstep.kind != "gett" or any(dstep.pre) or any(jstep.pre) or jres not in (da, db) or da == db:
            return
        t = db if da == jres else da
        g, gd = jstep.spec, dstep.spec
        if gd.B * gd.M * gd.N != 1 or self.layout[t] != self.layout[jres] or len(g.k) != 1 or g.K < 64 \
                or g.B * g.M * g.N < (1 << 20) or g.b:
            return
        isz = self.dtype.itemsize
        self.plan[-2:] = [("pairdot", ja, jb, t, dres, jstep, dstep, jres, da == jres)]
        self.info[-2:] = [StepInfo("gett", jstep.mults + dstep.mults,
                                   isz * g.B * (g.M * g.K + g.K * g.N + g.M * g.N), (g.B, g.M, g.N, g.K), False)]

    @staticmethod
    def _entry_io(entry):
        """(operand ssa ids, result ssa id) of a plan entry"""
        k = entry[0]
        if k == "single":
            return (entry[1],), entry[2]
        if k == "chain2":
            return (entry[1], entry[2], entry[3]), entry[4]
        if k == "rowpass":
            return (() if entry[1] is None else (entry[1],)) + tuple(entry[2]), entry[3]
        if k == "pairdot":
            return (entry[1], entry[2], entry[3]), entry[4]
        return (entry[1], entry[2]), entry[3]

    def _assign_lanes(self, max_lanes=8, min_steps=4):
        """Independent branches of the tree run concurrently: every plan entry gets a LANE (a HIP stream at run
        time).  A step with ONE sizeable operand sub-tree continues that sub-tree's chain on the same lane (a
        boundary sweep is one lane).  A JOIN -- two or more operand sub-trees of at least ``min_steps`` launches --
        stays on its parent's lane (the root's is lane 0, the caller's stream) and every chain below it opens a lane
        of its own, joining back with one event right before the join; joins below joins stay on the same lane, so
        the big GEMM-shaped joins of ``quadrant_path_2d`` run one after the other on lane 0 while its four corner
        sweeps -- many small, latency-bound launches -- overlap on lanes 1..4."""
        n = len(self.plan)
        prod = {}
        for i, e in enumerate(self.plan):
            prod[self._entry_io(e)[1]] = i
        count, kids = [1] * n, [()] * n
        for i, e in enumerate(self.plan):
            ch = tuple(prod[o] for o in self._entry_io(e)[0] if o in prod)
            kids[i] = ch
            count[i] = 1 + sum(count[c] for c in ch)
        lane = [0] * n
        nxt = [1]
        home_taken = [False]
        big_kids = lambda i: [c for c in kids[i] if count[c] >= min_steps]
        stack = [(n - 1, 0)] if n else []
        while stack:
            i, ln = stack.pop()
            lane[i] = ln
            big = big_kids(i)
            for c in kids[i]:
                if c not in big:
                    stack.append((c, ln))
            if len(big) >= 2:
                # a stack: pushed last = handled first = takes the home lane.  Join order: the FIRST join's first chain
                # (that join then sits right behind its own operand on lane 0).  Plain order: the LAST join's chain --
                # lane 0 finishes it before it reaches the first join, so no join runs beside a corner sweep
                for c in (reversed(big) if self.join_order else big):
                    if len(big_kids(c)) >= 2 or nxt[0] >= max_lanes:
                        stack.append((c, ln))          # a join below a join / out of lanes: same lane
                    elif ln == 0 and not home_taken[0] and not (self.join_order and any(len(big_kids(c_)) >= 2 for c_ in big)):
                        home_taken[0] = True           # ONE chain runs ahead of the joins on the caller's stream:
                        stack.append((c, 0))           # four corner sweeps = four streams = the default HW queues
                    else:
                        stack.append((c, nxt[0]))
                        nxt[0] += 1
            elif big:
                stack.append((big[0], ln))
        self.lanes = lane
        self.nlanes = nxt[0]
        self._producer = prod
        self._order_for_joins()

    def _order_for_joins(self):
        """Issue order and lane priorities of a laned plan.  The ANCHORS are the entries of lane 0 that take an operand
        from another lane (the joins).  Everything the first anchor needs is issued first -- its chains interleaved
        launch by launch, so that they advance together however slow the host is -- then the anchor itself, then what the
        second anchor needs, and so on: the first join starts as soon as ITS operands exist and the chains of the later
        joins run beside it instead of in front of it.  Any topological order is a valid schedule; this one only
        changes WHEN things are enqueued (``lane_priority``: HIP stream priority per lane, all normal by default)."""
        n = len(self.plan)
        self.lane_priority = [0] * self.nlanes
        if self.nlanes <= 1 or not self.options.join_order:
            return
        if not self.join_order:
            self._interleave_chains()
            return
        prod = self._producer
        kids = [tuple(prod[o] for o in self._entry_io(e)[0] if o in prod) for e in self.plan]
        anchors = [i for i in range(n) if self.lanes[i] == 0 and any(self.lanes[c] != 0 for c in kids[i])]
        if not anchors:
            return
        if anchors[-1] != n - 1:
            anchors.append(n - 1)
        need = [None] * n
        for a in anchors:
            stack = [a]
            while stack:
                i = stack.pop()
                if need[i] is not None:
                    continue
                need[i] = a
                stack.extend(kids[i])
        order = []
        for a in anchors:
            group = [i for i in range(n) if need[i] == a and i != a]
            queues = {}
            for i in group:
                queues.setdefault(self.lanes[i], []).append(i)
            qs = [queues[l] for l in sorted(queues)]
            issued = set(order)
            while any(qs):
                progressed = False
                for q in qs:
                    if q and all(c in issued for c in kids[q[0]]):
                        i = q.pop(0)
                        order.append(i)
                        issued.add(i)
                        progressed = True
                if not progressed:       # (cannot happen for a valid plan; keep the original order rather than spin)
                    rest = sorted(i for q in qs for i in q)
                    order.extend(rest)
                    break
            order.append(a)
        order += [i for i in range(n) if need[i] is None]
        if sorted(order) != list(range(n)):
            return
        # (HIP stream priorities for the first join's lanes were tried in round 4 and dropped: they did not keep a join
        # from slowing down beside a corner sweep, and streams of a second priority class cost hardware queues.
        # options.lane_priority = True brings them back for experiments.)
        if self.options.lane_priority:
            first = anchors[0]
            for i in range(n):
                if need[i] == first:
                    self.lane_priority[self.lanes[i]] = -1
        # HOLD: the chains of the LATER joins wait until the first join's own chains are done, i.e. they run beside the
        # first join, not beside its chains.  Beside a join a sweep crawls (an eighth of its speed: the join holds every
        # CU's LDS and matrix pipe) and ends shortly after it, so this pays only when the join is long: the 7 ms joins of the
        # whole 10x10 D=6 network (4.96 rounds of tiles) gain 0.2 ms per step (two corner sweeps in 1.03 ms instead of four
        # in 1.85 ms, the join 0.25 ms slower), the 3.5 / 1.8 / 0.95 ms joins of a rank of 2 / 4 / 8 LOSE 0.2-0.35 ms
        # (measured, round 4).  Round 6: with every row of a corner one MFMA-bound launch (rowq.hip: a corner is 5 launches
        # and ~0.35 ms instead of 0.8) a held chain no longer gets onto a CU before the join's tail -- its first launch waits
        # out the whole join -- and all four corners up front cost less than the hold ever hid: 15.8 ms held, 14.7 ms not
        # (profiles/r06_hold_late.txt).  "auto" therefore holds only plans WITHOUT fused rows; "0" / "1" force it off / on.
        self.hold_late = None
        hold = self.options.hold_late
        fused_rows = any(e[0] == "rowpass" for e in self.plan)
        if len(anchors) > 1 and hold != "0" and (hold == "1" or (self.info[anchors[0]].mults >= HOLD_LATE_MIN_MULTS
                                                                  and not fused_rows)):
            first = anchors[0]
            early = sorted({self.lanes[i] for i in range(n) if need[i] == first and i != first})
            late = sorted({self.lanes[i] for i in range(n) if need[i] is not None and need[i] != first} - set(early) - {0})
            if late:
                self.hold_late = (order.index(first), late, early)      # (position of the first join in the issue order, ...)
        self.plan = [self.plan[i] for i in order]
        self.info = [self.info[i] for i in order]
        self.lanes = [self.lanes[i] for i in order]
        self._producer = {self._entry_io(e)[1]: i for i, e in enumerate(self.plan)}

    def _interleave_chains(self):
        """Plain order of a laned plan (what a launch program records): the entries of the side chains and of lane 0's
        own chain dealt round robin, lane 0's joins behind them.  Nothing overlaps that did not before -- every chain
        still ends before the first join starts -- but a step that begins on an idle device (a rank whose previous step
        ended in a collective) has all its chains started within the first few launches instead of one chain's 22
        launches after the other."""
        n = len(self.plan)
        prod = self._producer
        kids = [tuple(prod[o] for o in self._entry_io(e)[0] if o in prod) for e in self.plan]
        first_join = next((i for i in range(n) if self.lanes[i] == 0 and any(self.lanes[c] != 0 for c in kids[i])), n)
        queues = {}
        for i in range(first_join):
            queues.setdefault(self.lanes[i], []).append(i)
        qs = [queues[l] for l in sorted(queues)]
        order, issued = [], set()
        while any(qs):
            progressed = False
            for q in qs:
                if q and all(c in issued for c in kids[q[0]]):
                    issued.add(q[0])
                    order.append(q.pop(0))
                    progressed = True
            if not progressed:
                return                      # (not a chain structure after all: keep the plan order)
        order += list(range(first_join, n))
        if sorted(order) != list(range(n)):
            return
        self.plan = [self.plan[i] for i in order]
        self.info = [self.info[i] for i in order]
        self.lanes = [self.lanes[i] for i in order]
        self._producer = {self._entry_io(e)[1]: i for i, e in enumerate(self.plan)}

    # ---- accounting -------------------------------------------------------------
    def flops(self, per_slice=False, hoist=True):
        """Floating-point operations actually executed (2 per real multiply-add,
        8 per complex).  With ``hoist`` slice-independent steps count once."""
        f = 8 if self.dtype.kind == "c" else 2
        ns = 1 if per_slice else self.tree.nslices
        tot = 0
        for inf in self.info:
            rep = ns if (inf.sliced_dep or not hoist) else 1
            tot += f * inf.mults * rep
        return tot

    def algorithmic_bytes(self, hoist=True):
        ns = self.tree.nslices
        return sum(inf.bytes * (ns if (inf.sliced_dep or not hoist) else 1) for inf in self.info)

    # ---- execution ----------------------------------------------------------------
    def _slice_values(self, s):
        vals = {}
        for ix in reversed(self.tree.sliced_inds):
            d = self.tree.size_dict[ix]
            vals[ix] = s % d
            s //= d
        return vals

    def _slice_input(self, x, i, vals):
        axes = self.input_sliced_axes[i]
        if not axes:
            return x
        key = [slice(None)] * x.ndim
        for ax, ix in axes:
            key[ax] = vals[ix]
        return x[tuple(key)]

    def _run_core(self, inputs, exponent, cache, only_independent=False, lanes=False):
        dev = inputs[0]._dev
        # (a launch-program recording never touches torch's streams)
        home = dev.torch.cuda.current_stream(dev.tdev) if hasattr(dev, "torch") and getattr(dev, "record", None) is None else None
        # the kernel pins this executor CAPTURED WHEN BUILT (options.pair_kernel / tile_cfg / split_k), not whatever the
        # thread's options are at call time (quimb_amd/options.py; the plan interpreter of the CPU tests has no pins)
        pinned = dev.pinned(self.options) if hasattr(dev, "pinned") else contextlib.nullcontext()
        try:
            with pinned:
                return self._run_core_impl(inputs, exponent, cache, only_independent, lanes)
        finally:
            if home is not None:
                dev.torch.cuda.set_stream(home)      # lanes switch the current stream: always hand the caller's back

    def _run_core_impl(self, inputs, exponent, cache, only_independent=False, lanes=False):
        """Evaluate the tree for one slice.  ``cache`` maps ssa id -> Array for
        slice-independent intermediates (filled on first use).

        With ``exponent`` (strip_exponent) every GETT step runs the fused epilogue:
        it scales by 1/(max|a| max|b|) read from the operands' absmax slots and
        reduces max|out| into its own slots; nothing is re-read to normalise.  The
        log10 of every max is summed on the device at the end."""
        dev = inputs[0]._dev
        live = dict(enumerate(inputs))
        slots = None
        if exponent is not None:
            nid = len(inputs) + len(self.tree.steps)
            slots = dev.new_slots(nid, self.dtype)
            has_scale = set()
        # branch concurrency (unsliced runs on the HIP device): lane -> stream; lane 0 is the caller's stream.  While a
        # launch program is being recorded (quimb_amd/program.py) the lanes are the PROGRAM's: switches and waits are
        # appended to it instead of acting on torch streams, and buffers come from (and go back to) its pool.
        rec = getattr(dev, "record", None)
        streams = None
        use_lanes = lanes and self.nlanes > 1 and self.options.lanes
        if rec is not None:
            if use_lanes:
                for l_ in range(1, self.nlanes):
                    rec.wait(l_, 0)                   # the slots / exponent fills recorded so far sit on lane 0
        elif use_lanes and hasattr(dev, "lane_streams"):
            streams = dev.lane_streams(self.nlanes, getattr(self, "lane_priority", None))
            for st_ in streams[1:]:
                st_.wait_stream(streams[0])       # inputs / slots are ready on the caller's stream
        laned = streams is not None or (rec is not None and use_lanes)
        keep_alive = []   # buffers handed from one lane to another stay allocated until every launch is queued
        foreign = set()   # ssa ids of such buffers: a program's pool never reuses them
        # options.lane_trace (debugging aid): HIP events at the first and last launch of every lane -> self.lane_trace
        trace = trace_at = None
        if streams is not None and self.options.lane_trace:
            trace = self.lane_trace = []
            first, last = {}, {}
            for i_, l_ in enumerate(self.lanes):
                first.setdefault(l_, i_)
                last[l_] = i_
            trace_at = set(first.values()) | set(last.values()) | {i_ for i_, l_ in enumerate(self.lanes) if l_ == 0 and i_ >= len(self.lanes) - 3}
        uses = {}
        def operands(entry):
            if entry[0] == "single":
                return (entry[1],)
            if entry[0] == "chain2":
                return (entry[1], entry[2], entry[3])
            if entry[0] == "rowpass":
                return (() if entry[1] is None else (entry[1],)) + tuple(entry[2])
            if entry[0] == "pairdot":
                return (entry[1], entry[2], entry[3])
            return (entry[1], entry[2])

        for entry in self.plan:
            for s in operands(entry):
                uses[s] = uses.get(s, 0) + 1
        def run_pair(a, b, res, step):
            """one pairwise step on the live operands (with the fused exponent epilogue where the kernels have one)"""
            if exponent is not None and step.kind == "gett" and self.dtype.kind != "c":
                ep = (
                    dev.slots_row(slots, a) if a in has_scale else None,
                    dev.slots_row(slots, b) if b in has_scale else None,
                    dev.slots_row(slots, res),
                )
                x = run_pair_step(step, live[a], live[b], ep=ep)
                has_scale.add(res)
                return x
            xa, xb = live[a], live[b]
            if exponent is not None:
                # operands with a pending (deferred) division: apply it now
                for s_, x_ in ((a, xa), (b, xb)):
                    if s_ in has_scale:
                        dev.div_by_absmax(x_._buf, x_.size, dev.slots_row(slots, s_), x_.dtype)
            x = run_pair_step(step, xa, xb)
            if exponent is not None and x.size:
                dev.strip_exponent(x._buf, x.size, x.dtype, exponent)
            return x

        held = getattr(self, "hold_late", None)
        for pi, entry in enumerate(self.plan):
            if laned and held is not None and pi == held[0]:
                # right before the first join is issued: the later joins' chains wait for what its chains have been
                # given so far (= all of them), not for the join itself
                for late_ in held[1]:
                    for early_ in held[2]:
                        if rec is not None:
                            rec.wait(late_, early_)
                        else:
                            streams[late_].wait_stream(streams[early_])
            if laned:
                for o in self._entry_io(entry)[0]:
                    pj = self._producer.get(o)
                    if pj is not None and self.lanes[pj] != self.lanes[pi]:
                        keep_alive.append(live[o])
                        foreign.add(o)
                        if rec is not None:
                            rec.wait(self.lanes[pi], self.lanes[pj])          # the join: one event at replay
                            continue
                        mine = streams[self.lanes[pi]]
                        mine.wait_stream(streams[self.lanes[pj]])     # the join: one event
                        buf_ = getattr(live[o], "_buf", None)
                        if hasattr(buf_, "record_stream") and not dev.torch.cuda.is_current_stream_capturing():
                            buf_.record_stream(mine)      # allocated on the producer's stream, read on this one: the
                                                          # allocator must not hand it out again before this stream is done
                if rec is not None:
                    rec.set_lane(self.lanes[pi])
            if streams is not None:
                dev.torch.cuda.set_stream(streams[self.lanes[pi]])
                if trace is not None and pi in trace_at:
                    ev = dev.torch.cuda.Event(enable_timing=True)
                    ev.record()
                    trace.append((f"lane {self.lanes[pi]} entry {pi} ({entry[0]}) start", ev))
            if entry[0] == "single":
                _, a, res, src, out = entry
                independent = not self.dep[res]
                if independent and cache is not None and res in cache:
                    live[res] = cache[res]
                elif only_independent and not independent:
                    continue
                else:
                    if exponent is not None and a in has_scale:
                        # apply the operand's deferred division before the single-term op
                        dev.div_by_absmax(live[a]._buf, live[a].size, dev.slots_row(slots, a), live[a].dtype)
                    live[res] = _einsum_single(live[a], src, out)
                    if independent and cache is not None:
                        cache[res] = live[res]
                ids = (a,)
            elif entry[0] == "chain2":
                _, a, w1, w2, res, c2 = entry
                independent = not self.dep[res]
                if independent and cache is not None and res in cache:
                    live[res] = cache[res]
                elif only_independent and not independent:
                    continue
                else:
                    x = Array.empty(c2.out_shape, self.dtype, dev)
                    ep = None
                    if exponent is not None:
                        ep = tuple(dev.slots_row(slots, s_) if s_ in has_scale else None for s_ in (a, w1, w2))
                        ep = ep + (dev.slots_row(slots, res),)
                        has_scale.add(res)
                    if self.options.chain2_kernel != "auto":
                        dev.contract_chain2(c2, self.dtype, live[a]._buf, live[w1]._buf, live[w2]._buf, x._buf, ep,
                                            pin=self.options.chain2_kernel)
                    else:
                        dev.contract_chain2(c2, self.dtype, live[a]._buf, live[w1]._buf, live[w2]._buf, x._buf, ep)
                    live[res] = x
                    if independent and cache is not None:
                        cache[res] = x
                ids = (a, w1, w2)
            elif entry[0] == "rowpass":
                _, a, wids, res, rp = entry
                independent = not self.dep[res]
                if independent and cache is not None and res in cache:
                    live[res] = cache[res]
                elif only_independent and not independent:
                    continue
                else:
                    x = Array.empty(rp.out_shape, self.dtype, dev)
                    ep = None
                    if exponent is not None:
                        ep = tuple(dev.slots_row(slots, s_) if (s_ is not None and s_ in has_scale) else None
                                   for s_ in (a,) + tuple(wids))
                        ep = ep + (dev.slots_row(slots, res),)
                        has_scale.add(res)
                    dev.contract_rowpass(rp, self.dtype, None if a is None else live[a]._buf,
                                         [live[w_]._buf for w_ in wids], x._buf, ep)
                    live[res] = x
                    if independent and cache is not None:
                        cache[res] = x
                ids = (() if a is None else (a,)) + tuple(wids)
            elif entry[0] == "pairdot":
                _, a, b, t, res, jstep, dstep, jres, join_first = entry
                x = Array.empty(dstep.out_shape, self.dtype, dev)
                ka, kb = (b, a) if jstep.swapped else (a, b)
                ep = None
                if exponent is not None:
                    ep = tuple(dev.slots_row(slots, s_) if s_ in has_scale else None for s_ in (ka, kb, t))
                    ep = ep + (dev.slots_row(slots, res),)
                fused = getattr(dev, "contract_pair_dot", None)
                if fused is not None and fused(jstep.spec, self.dtype, live[ka]._buf, live[kb]._buf, live[t]._buf, x._buf, ep):
                    if exponent is not None:
                        has_scale.add(res)
                else:           # not a contraction the device fuses: the two steps it was
                    live[jres] = run_pair(a, b, jres, jstep)
                    x = run_pair(*((jres, t) if join_first else (t, jres)), res, dstep)
                    gone = live.pop(jres, None)
                    if rec is not None and gone is not None:
                        rec.release(gone._buf)
                live[res] = x
                ids = (a, b, t)
            else:
                _, a, b, res, step = entry
                independent = not self.dep[res]
                if independent and cache is not None and res in cache:
                    live[res] = cache[res]
                elif only_independent and not independent:
                    continue
                else:
                    x = run_pair(a, b, res, step)
                    live[res] = x
                    if independent and cache is not None:
                        cache[res] = x
                ids = (a, b)
            for s in ids:
                uses[s] -= 1
                if uses[s] == 0:
                    gone = live.pop(s, None)
                    # a recorded program reuses the block for later launches of THIS lane (stream order); inputs and
                    # buffers that crossed lanes are left alone
                    if rec is not None and gone is not None and s >= len(inputs) and s not in foreign:
                        rec.release(gone._buf)
            if streams is not None and trace is not None and pi in trace_at:
                ev = dev.torch.cuda.Event(enable_timing=True)
                ev.record()
                trace.append((f"lane {self.lanes[pi]} entry {pi} ({entry[0]}) end", ev))
        if streams is not None:
            dev.torch.cuda.set_stream(streams[0])
        if rec is not None and laned:
            rec.set_lane(0)
        out = live.get(self.root)
        if exponent is not None and out is not None:
            if self.root in has_scale:
                dev.div_by_absmax(out._buf, out.size, dev.slots_row(slots, self.root), out.dtype)
            dev.slots_log10_sum(slots, self.dtype, exponent)
        return out

    def program(self, arrays, strip_exponent=False, mark_min_mults=None):
        """Record one whole (unsliced) contraction as a LAUNCH PROGRAM and return it (``quimb_amd.program``):
        ``p()`` / ``p(arrays)`` replays the recorded launches -- branches on their own HIP streams -- with one host call
        and no per-step Python; inputs are read in place (re-based pointers, no copies).  ``mark_min_mults``: launches
        of at least that many multiplications carry timing events (``p(timing_slot=k)``, ``p.timings(k)``)."""
        from .array import asarray as _asarray
        from .program import ContractionProgram, EagerProgram

        if not hasattr(_asarray(arrays[0])._dev, "lib"):       # the plan interpreter of the CPU tests: nothing to record
            return EagerProgram(self, arrays, strip_exponent, mark_min_mults)
        # MFMA-bound joins and HBM-bound corner sweeps do NOT overlap well on this chip (measured, round 4: a 0.95 ms
        # join takes 1.8 ms beside a corner sweep -- the sweeps evict its operand panels from the L2), and a program's
        # host side is no bottleneck: record the plain order -- every chain side by side, then the joins
        ex = self
        if self.join_order and self.nlanes > 1 and getattr(self, "hold_late", None) is None \
                and not self.options.program_join_order:
            ex = getattr(self, "_plain_order_twin", None)
            if ex is None:
                ex = self._plain_order_twin = TreeExecutor(self.tree, self.dtype, join_order=False, options=self.options)
        return ContractionProgram(ex, arrays, strip_exponent, mark_min_mults)

    def check_inputs(self, arrays):
        """The arrays as device arrays of the plan's dtype -- after checking that there is one per input of the tree, of
        the size the tree says.  EVERY way into ``_run_core`` goes through here: a plan run on arrays of other shapes
        launches kernels on addresses those arrays do not have."""
        tree = self.tree
        if len(arrays) != len(tree.inputs):
            raise ValueError(f"expected {len(tree.inputs)} arrays, got {len(arrays)}")
        xs = [asarray(x).astype(self.dtype) for x in arrays]
        for x, t in zip(xs, tree.inputs):
            want = tuple(tree.size_dict[ix] for ix in t)
            if x.shape != want:
                raise ValueError(f"array shape {x.shape} does not match indices {t} with sizes {want}")
        return xs

    def graph(self, arrays, strip_exponent=False):
        """Capture one whole (unsliced) contraction into a HIP graph and return a
        ``GraphedContraction``: ``g.replay()`` re-launches the recorded kernel sequence
        with one host call (dispatch-bound networks: circuits, DMRG matvecs).  The inputs
        are static device buffers -- refresh them in place with ``g.update(i, array)``."""
        return GraphedContraction(self, arrays, strip_exponent)

    def _run_slices_graphed(self, xs, todo):
        """strip_exponent slice loop as hipGraph replays (HIP device only)."""
        dev = xs[0]._dev
        torch = dev.torch
        sliced = [i for i in range(len(xs)) if self.input_sliced_axes[i]]
        key = ("slice_graph", tuple(x.shape for x in xs))
        ent = getattr(self, "_slice_graphs", {}).get(key)
        if ent is None:
            vals0 = self._slice_values(todo[0])
            static = [self._slice_input(x, i, vals0) for i, x in enumerate(xs)]
            static = [s.copy() if s is x else s for s, x in zip(static, xs)]  # private buffers for the graph
            side = torch.cuda.Stream(device=dev.tdev)
            side.wait_stream(torch.cuda.current_stream(dev.tdev))
            with torch.cuda.stream(side):                                      # warm-up: plans, tables, allocator
                for _ in range(2):
                    self._run_core(static, dev.new_exponent(), None)
            torch.cuda.current_stream(dev.tdev).wait_stream(side)
            torch.cuda.synchronize(dev.tdev)
            out_shape = tuple(self.tree.size_dict[ix] for ix in self.tree.output)
            acc = Array.full(out_shape, 0.0, self.dtype, dev)
            g_exp = dev.new_exponent()
            acc_exp = dev.new_exponent_neg_inf()
            torch.cuda.synchronize(dev.tdev)
            g = torch.cuda.CUDAGraph()
            # thread-local capture: other threads of the process (RCCL's watchdog polling its events in a
            # multi-GPU job) must not invalidate the capture
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                g_exp.zero_()
                out = self._run_core(static, g_exp, None)
                dev.axpby_exp(acc._buf, out._buf, acc.size, acc_exp, g_exp, acc.dtype)
            ent = dict(static=static, g=g, acc=acc, acc_exp=acc_exp, keep=(out, g_exp))
            if not hasattr(self, "_slice_graphs"):
                self._slice_graphs = {}
            self._slice_graphs[key] = ent
        static, acc, acc_exp = ent["static"], ent["acc"], ent["acc_exp"]
        dev.fill(acc._buf, acc.size, 0.0, acc.dtype)
        acc_exp.fill_(float("-inf"))
        for i in range(len(xs)):                                               # unsliced inputs: refresh once
            if i not in sliced:
                static[i]._buf[: xs[i].size].copy_(xs[i]._buf[: xs[i].size])
        for s in todo:
            vals = self._slice_values(s)
            for i in sliced:
                shape, strides, offset = _slice_view(self, xs[i], i, vals)
                dev.permute(static[i]._buf, xs[i]._buf, shape, strides, offset, self.dtype)
            ent["g"].replay()
        return acc.copy(), dev.read_exponent(acc_exp)

    def __call__(self, arrays, strip_exponent=False, slices=None, hoist=True, defer_exponent=False, lanes=None,
                 slice_graph=None):
        """Contract.  ``slices``: iterable of slice numbers to evaluate (default
        all); the partial sum over exactly those slices is returned, which is what
        a rank of the multi-GPU driver needs before the RCCL reduce.

        Returns an ``Array`` (or ``(Array, exponent)`` if ``strip_exponent``).  ``defer_exponent`` (unsliced trees):
        the exponent comes back as the device-resident accumulator instead of a float, so the call does not
        synchronise with the device -- ``dev.read_exponent(e)`` reads it later.  ``lanes`` / ``slice_graph``: per-call
        overrides of the executor's options (False: every launch on the caller's stream / slices launch by launch)."""
        tree = self.tree
        xs = self.check_inputs(arrays)
        dev = xs[0]._dev
        nsl = tree.nslices
        if slices is not None:
            slices = [int(s) for s in slices]
            if any(s < 0 or s >= nsl for s in slices):
                raise ValueError(f"slice numbers must lie in [0, {nsl}), got {slices}")
            if len(set(slices)) != len(slices):
                raise ValueError(f"duplicate slice numbers in {slices}")
            if nsl == 1 and not slices:
                # an unsliced tree is ONE slice (number 0): a rank that does not own it contributes zero,
                # with exponent -inf, exactly like a rank without slices of a sliced tree
                zero = Array.full([tree.size_dict[ix] for ix in tree.output], 0.0, self.dtype, dev)
                return (zero, float("-inf")) if strip_exponent else zero
        if not tree.steps:
            out = _einsum_single(xs[0], self.input_inds[0], tuple(tree.output))
            return (out, 0.0) if strip_exponent else out
        if nsl == 1:
            exponent = dev.new_exponent() if strip_exponent else None
            out = self._run_core(xs, exponent, None, lanes=True if lanes is None else bool(lanes))
            if strip_exponent:
                return (out, exponent) if defer_exponent else (out, dev.read_exponent(exponent))
            return out

        todo = range(nsl) if slices is None else list(slices)
        cache = {} if (hoist and not strip_exponent) else None
        acc, acc_e = None, None
        acc_exp = None   # device-resident exponent of the running sum: the slice loop never syncs with the host
        if strip_exponent and len(todo) >= 4 and hasattr(dev, "torch") \
                and (self.options.slice_graph if slice_graph is None else bool(slice_graph)):
            # every slice runs the SAME launch sequence on differently sliced inputs: record it once as a
            # hipGraph over static input buffers and replay it per slice (a few tiny slicing copies + one
            # graph launch instead of ~80 Python-driven launches)
            if not getattr(self, "_slice_graph_broken", False):
                try:
                    return self._run_slices_graphed(xs, todo)
                except RuntimeError as err:      # capture refused (driver / allocator state): plain loop from now on
                    import warnings

                    self._slice_graph_broken = True
                    warnings.warn(f"quimb_amd: hipGraph capture of the slice loop failed ({err}); "
                                  "falling back to launch-by-launch slices")
        for s in todo:
            vals = self._slice_values(s)
            ins = [self._slice_input(x, i, vals) for i, x in enumerate(xs)]
            if strip_exponent:
                exponent = dev.new_exponent()
                out = self._run_core(ins, exponent, None)
                if acc is None:
                    acc = Array.full(out.shape, 0.0, out.dtype, dev)
                    acc_exp = dev.new_exponent_neg_inf()
                dev.axpby_exp(acc._buf, out._buf, acc.size, acc_exp, exponent, acc.dtype)
            else:
                out = self._run_core(ins, None, cache)
                if acc is None:
                    acc = out.copy() if (cache is not None and self.root in cache) else out
                else:
                    dev.axpby(acc._buf, out._buf, acc.size, 1.0, 1.0, acc.dtype)
        if acc is None:  # this rank owns no slices
            acc = Array.full([tree.size_dict[ix] for ix in tree.output], 0.0, self.dtype, dev)
            acc_e = float("-inf") if strip_exponent else None
        elif strip_exponent:
            acc_e = dev.read_exponent(acc_exp)   # the one read-back of the whole slice loop
        return (acc, acc_e) if strip_exponent else acc


def _slice_view(ex, x, i, vals):
    """(shape, strides, offset) of input ``i`` sliced at ``vals`` -- the view ``_slice_input`` copies."""
    from .pairwise import contig_strides

    st = contig_strides(x.shape)
    fixed = {ax: vals[ix] for ax, ix in ex.input_sliced_axes[i]}
    shape = [d for ax, d in enumerate(x.shape) if ax not in fixed]
    strides = [s for ax, s in enumerate(st) if ax not in fixed]
    offset = sum(v * st[ax] for ax, v in fixed.items())
    return shape, strides, offset


class GraphedContraction:
    """A contraction recorded as a hipGraph (via torch's stream capture -- plumbing; every
    node is one of this library's HIP kernels)."""

    def __init__(self, executor, arrays, strip_exponent=False):
        if executor.tree.nslices != 1:
            raise ValueError("graph capture supports unsliced trees")
        self.executor = executor
        self.strip_exponent = strip_exponent
        self.inputs = [x.copy() for x in executor.check_inputs(arrays)]
        dev = self.inputs[0]._dev
        if not hasattr(dev, "torch"):
            raise RuntimeError("graph capture needs the HIP device")
        torch = dev.torch
        self._dev = dev
        # warm-up on a side stream: compiles every plan / table before capture
        s = torch.cuda.Stream(device=dev.tdev)
        s.wait_stream(torch.cuda.current_stream(dev.tdev))
        with torch.cuda.stream(s):
            for _ in range(2):
                executor._run_core(self.inputs, dev.new_exponent() if strip_exponent else None, None, lanes=True)
        torch.cuda.current_stream(dev.tdev).wait_stream(s)
        torch.cuda.synchronize(dev.tdev)
        self._graph = torch.cuda.CUDAGraph()
        self._exponent = dev.new_exponent() if strip_exponent else None
        with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
            if strip_exponent:
                self._exponent.zero_()
            self.output = executor._run_core(self.inputs, self._exponent, None, lanes=True)

    def update(self, i, array):
        """Overwrite static input ``i`` in place (device-to-device or host-to-device copy)."""
        src = asarray(array).astype(self.executor.dtype)
        if src.shape != self.inputs[i].shape:
            raise ValueError("shape mismatch")
        self.inputs[i]._buf[: src.size].copy_(src._buf[: src.size])

    def replay(self, defer_exponent=False):
        """One host call re-launches the recorded kernel sequence (independent branches as parallel graph branches).
        ``defer_exponent``: hand back the device-resident exponent accumulator instead of reading it (no sync)."""
        self._graph.replay()
        if self.strip_exponent:
            return self.output, (self._exponent if defer_exponent else self._dev.read_exponent(self._exponent))
        return self.output
