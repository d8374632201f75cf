"""Contraction-order search (host, pure python).

In the reference this is cotengra's job (``optimize="greedy"`` is quimb's
default strategy, quimb/tensor/contraction.py:11; structured 2D networks use
explicit boundary sweeps, quimb/tensor/tn2d/core.py:1355-1484).  cotengra is not
part of this build, so the executor ships its own finders:

* ``greedy_path``      -- size-difference greedy (the classic opt_einsum rule)
* ``random_greedy``    -- Boltzmann-perturbed greedy restarts, keep the cheapest
* ``sweep_path_2d``    -- row-by-row, site-by-site boundary sweep of an Lx x Ly grid
* ``quadrant_path_2d`` -- four corner sweeps + two GEMM-shaped joins + a dot product (MFMA-bound, shardable)
* ``bisection_ssa``    -- recursive balanced bisection + reconfigured leaves (finds the quadrant tree of a lattice)
* ``find_slices``      -- greedy choice of sliced indices (what cotengra's
                          SliceFinder does) so that the slices can be sharded
* ``modeled_time``     -- roofline time of the executor's plan for a tree; ``minimize="time"`` /
                          ``optimize="auto-time"`` make the finders intensity-aware
* ``set_tree_cache``   -- on-disk cache of found trees keyed by ``geometry_hash``
"""

import heapq
import json
import math
import os
import random

from .pairwise import prod
from .tree import ContractionTree, ssa_to_linear


def _result_inds(a, b, counts, out_set):
    seen = {}
    for ix in a:
        seen[ix] = 1
    for ix in b:
        seen[ix] = seen.get(ix, 0) + 1
    keep = []
    for ix in a + tuple(i for i in b if i not in a):
        if ix in out_set or counts[ix] > seen[ix]:
            keep.append(ix)
    return tuple(keep), seen


def greedy_ssa(inputs, output, size_dict, temperature=0.0, rng=None, costmod=1.0):
    """Greedy pairwise ordering.  score = size(out) - costmod*(size(a)+size(b)),
    optionally Boltzmann-perturbed (``temperature`` > 0)."""
    inputs = [tuple(dict.fromkeys(t)) for t in inputs]
    out_set = set(output)
    n = len(inputs)
    terms = {i: t for i, t in enumerate(inputs)}
    counts = {}
    where = {}
    for i, t in terms.items():
        for ix in t:
            counts[ix] = counts.get(ix, 0) + 1
            where.setdefault(ix, set()).add(i)
    tsize = {i: prod(size_dict[ix] for ix in t) for i, t in terms.items()}
    rng = rng or random

    def score(i, j):
        keep, _ = _result_inds(terms[i], terms[j], counts, out_set)
        so = prod(size_dict[ix] for ix in keep)
        s = so - costmod * (tsize[i] + tsize[j])
        if temperature > 0:
            gumbel = -math.log(-math.log(rng.random() * 0.999999 + 1e-12))
            s = s - temperature * abs(s if s != 0 else 1.0) * gumbel
        return s

    heap = []
    for ix, ts in where.items():
        ts = sorted(ts)
        for a in range(len(ts)):
            for b in range(a + 1, len(ts)):
                heapq.heappush(heap, (score(ts[a], ts[b]), ts[a], ts[b]))
    ssa = []
    nxt = n
    while len(terms) > 1:
        pick = None
        while heap:
            s, i, j = heapq.heappop(heap)
            if i in terms and j in terms:
                pick = (i, j)
                break
        if pick is None:
            # disconnected components: outer products, smallest first
            i, j = sorted(terms, key=lambda t: tsize[t])[:2]
            pick = (i, j)
        i, j = pick
        keep, seen = _result_inds(terms[i], terms[j], counts, out_set)
        for t in (i, j):
            for ix in terms[t]:
                where[ix].discard(t)
        for ix, c in seen.items():
            counts[ix] -= c
        for ix in keep:
            counts[ix] += 1
            where[ix].add(nxt)
        del terms[i], terms[j]
        terms[nxt] = keep
        tsize[nxt] = prod(size_dict[ix] for ix in keep)
        ssa.append((i, j))
        neigh = set()
        for ix in keep:
            neigh |= where[ix]
        neigh.discard(nxt)
        for t in neigh:
            heapq.heappush(heap, (score(t, nxt), t, nxt))
        nxt += 1
    return ssa


def greedy_path(inputs, output, size_dict):
    n = len(inputs)
    return ssa_to_linear(greedy_ssa(inputs, output, size_dict), n)


def random_greedy(inputs, output, size_dict, repeats=32, seed=0, temperature=0.3, minimize="flops", dtype="float32"):
    """Best of ``repeats`` perturbed greedy runs: by total multiplications (``minimize="flops"``, what
    cotengra's default objective does) or by ``modeled_time`` (``"time"``: bytes moved and launch count matter
    as much as multiplications for the HBM-bound steps that dominate boundary-like networks)."""
    rng = random.Random(seed)
    best, best_cost = None, None
    for r in range(max(1, repeats)):
        ssa = greedy_ssa(
            inputs, output, size_dict,
            temperature=0.0 if r == 0 else temperature * rng.random(),
            rng=rng,
            costmod=1.0 if r == 0 else rng.choice([0.5, 1.0, 1.0, 2.0]),
        )
        tree = ContractionTree(inputs, output, size_dict, ssa_path=ssa)
        c = tree.contraction_cost() if minimize == "flops" else modeled_time(tree, dtype)
        if best_cost is None or c < best_cost:
            best, best_cost = tree, c
    return best


# gfx950 roofs the time model prices a step against (MI355X_MICROARCH.md: dense MFMA peaks, achievable HBM copy rate),
# and what this library's kernels reach of them (round-3 measurements, docs/history/DESIGN_rounds1-4.md section 5): the k-outer MFMA kernel
# 0.85 of the peak on GEMM-shaped fp32 joins, the older tiled kernels ~0.5; streaming / fused-pair kernels 4.9 of the
# 6.29 TB/s a copy reaches; a launch that moves less than a few MB is latency: ~12 us on the device
_PEAK_FLOPS = {"float32": 157.3e12, "float64": 78.6e12, "complex64": 157.3e12, "complex128": 78.6e12}
_HBM_BYTES_PER_S = 6.29e12
_HBM_EFF = 0.78
_MFMA_EFF_JOIN, _MFMA_EFF_TILED = 0.85, 0.5
_LAUNCH_S, _SMALL_LAUNCH_S, _SMALL_BYTES = 4e-6, 12e-6, 4 << 20


def _step_cost(inf, name, f, nslices):
    """(seconds, seconds of it that are HBM time) of one launch of the executor's plan, repeated per slice where the
    step depends on a sliced index -- the ONE place the time model prices a step."""
    rep = nslices if inf.sliced_dep else 1
    join = name == "float32" and inf.kind == "gett" and min(inf.M, inf.N) >= 128 and inf.K >= 64   # gemmk's domain
    t_mfma = f * inf.mults / (_PEAK_FLOPS[name] * (_MFMA_EFF_JOIN if join else _MFMA_EFF_TILED))
    t_hbm = inf.bytes / (_HBM_BYTES_PER_S * _HBM_EFF)
    t = max(t_mfma, t_hbm) + (_SMALL_LAUNCH_S if inf.bytes < _SMALL_BYTES else _LAUNCH_S)
    return rep * t, rep * (t_hbm if t_hbm >= t_mfma else 0.0)


def modeled_time(tree, dtype="float32"):
    """Estimate of executing ``tree`` on one MI355X: every launch of the plan the whole-tree executor would issue costs
    ``max(flops / (MFMA peak x kernel efficiency), algorithmic bytes / achieved HBM rate) + launch``, slices repeated,
    and the lanes of the plan (independent branches on their own HIP streams, ``TreeExecutor._assign_lanes``) overlap:
    the caller's lane runs in sequence, the side lanes cost the longest of them or their summed HBM time, whichever is
    larger -- the objective of the intensity-aware finders.  Fused pairs (two
    big-x-small steps in one pass, chain2q.hip) count with the bytes they actually move.  On the headline network it
    lands within 10 % of the device: site sweep 22.6 ms (measured 20.5), four-quadrant tree 14.8 ms (measured 16.0),
    the busiest rank's share of 2 / 4 / 8 ranks 7.6 / 3.9 / 2.2 ms (measured 9.1 / 5.4 / 3.6: the small and size-3
    steps of a share run further below the roofs than the model's one efficiency per kernel class says)."""
    import numpy as np

    from .executor import TreeExecutor   # planning only: no device is touched

    ex = TreeExecutor(tree, dtype)
    name = np.dtype(dtype).name
    f = 8 if np.dtype(dtype).kind == "c" else 2
    ns = tree.nslices
    lane_t, lane_hbm = {}, {}
    cost = [_step_cost(inf, name, f, ns) for inf in ex.info]      # (time, HBM-bound part) per launch, slices included
    for (t, hbm), lane in zip(cost, ex.lanes):
        lane_t[lane] = lane_t.get(lane, 0.0) + t
        lane_hbm[lane] = lane_hbm.get(lane, 0.0) + hbm
    side = [l for l in lane_t if l != 0]
    # lane 0: its own chain, then the joins; side lanes run beside lane 0's chain and share the HBM with it
    own = lane_t.get(0, 0.0)
    if not side:
        return own
    overlap = max(max(lane_t[l] for l in side), sum(lane_hbm[l] for l in side))
    # the part of lane 0 that precedes the first join is concurrent with the side lanes: count the larger of the two
    first_join = next((i for i, e in enumerate(ex.plan) if ex.lanes[i] == 0 and
                       any(ex.lanes[ex._producer[o]] != 0 for o in ex._entry_io(e)[0] if o in ex._producer)), len(ex.plan))
    pre = sum(cost[i][0] for i, lane in enumerate(ex.lanes) if lane == 0 and i < first_join)
    hbm_all = sum(lane_hbm[l] for l in side) + lane_hbm.get(0, 0.0) * (pre / own if own else 0.0)
    return (own - pre) + max(pre, overlap, hbm_all)


def fused_pair_count(tree, dtype="float32"):
    """Launches of the executor's plan that are fused pairs / triples of streaming steps."""
    from .executor import TreeExecutor

    return sum(1 for e in TreeExecutor(tree, dtype).plan if e[0] == "chain2")


def sweep_ssa_2d(Lx, Ly):
    """SSA path for a row-major Lx x Ly grid of tensors: absorb sites one at a
    time, left to right, top row to bottom row, into a single boundary tensor
    (the exact version of quimb's boundary contraction sweep,
    quimb/tensor/tn2d/core.py:1355-1484, without compression)."""
    n = Lx * Ly
    ssa, cur, nxt = [], 0, n
    for t in range(1, n):
        ssa.append((cur, t))
        cur = nxt
        nxt += 1
    return ssa


def sweep_path_2d(Lx, Ly):
    return ssa_to_linear(sweep_ssa_2d(Lx, Ly), Lx * Ly)


def quadrant_ssa_2d(Lx, Ly, rx=None, cy=None):
    """SSA path for a row-major Lx x Ly grid that contracts the four QUADRANTS (rows < rx / >= rx, columns < cy /
    >= cy, default the middle) site by site from their outer corners, joins the two upper and the two lower
    quadrants over the bonds they share and closes with the product of the two halves:

        T = TL . TR,   B = BL . BR,   Z = T . B

    On the 10 x 10 D = 6 lattice that is 9.84e11 multiplications (1.13 x the site sweep's 8.72e11) with a largest
    intermediate of 6^10 elements (242 MB instead of 1.45 GB), and ~96 % of them sit in the two joins -- plain
    7776^3 matrix products, MFMA-bound (AI ~1300 flop/B) where every step of the sweep is HBM-bound, and
    shardable by rows / columns of T and B with no redundancy (quimb_amd/quadrants.py).  The reference analogue is
    the tree cotengra finds for ``TensorNetwork.contraction_tree`` (quimb/tensor/tensor_core.py:9004-9014) and the
    closing product of a two-sided boundary contraction (quimb/tensor/tn2d/core.py:2493-2498)."""
    rx = Lx // 2 if rx is None else rx
    cy = Ly // 2 if cy is None else cy
    if not (0 < rx < Lx and 0 < cy < Ly):
        raise ValueError("the cut must leave four non-empty quadrants")
    nxt = [Lx * Ly]
    ssa = []
    sid = lambda r, c: r * Ly + c

    def sweep(sites):
        cur = sites[0]
        for s_ in sites[1:]:
            ssa.append((cur, s_))
            cur = nxt[0]
            nxt[0] += 1
        return cur

    tl = sweep([sid(r, c) for r in range(rx) for c in range(cy)])
    tr = sweep([sid(r, c) for r in range(rx) for c in range(Ly - 1, cy - 1, -1)])
    bl = sweep([sid(r, c) for r in range(Lx - 1, rx - 1, -1) for c in range(cy)])
    br = sweep([sid(r, c) for r in range(Lx - 1, rx - 1, -1) for c in range(Ly - 1, cy - 1, -1)])
    for pair in ((tl, tr), (bl, br)):
        ssa.append(pair)
        nxt[0] += 1
    ssa.append((nxt[0] - 2, nxt[0] - 1))
    return ssa


def quadrant_path_2d(Lx, Ly, rx=None, cy=None):
    return ssa_to_linear(quadrant_ssa_2d(Lx, Ly, rx, cy), Lx * Ly)


def geometry_hash(inputs, output, size_dict, extra=""):
    """Hash of a network's geometry that does not depend on the index NAMES: indices are renumbered by first
    appearance, exactly what ``TensorNetwork.geometry_hash`` keys its path caches on
    (quimb/tensor/tensor_core.py:5219-5293).  Two networks with the same hash accept the same ssa path."""
    import hashlib

    number = {}
    canon = [[number.setdefault(ix, len(number)) for ix in t] for t in inputs]
    out = [number[ix] for ix in output]
    sizes = [int(size_dict[ix]) for ix in number]          # insertion order == numbering order
    blob = repr((canon, out, sizes, extra)).encode()
    return hashlib.sha1(blob).hexdigest()


_TREE_CACHE_DIR = [None]


def set_tree_cache(directory):
    """Persist the trees found by the string-named strategies ("greedy", "random-greedy", "auto", "auto-hq")
    under ``directory``, keyed by ``geometry_hash`` + strategy -- the role cotengra's
    ``ReusableHyperOptimizer(directory=...)`` plays for quimb.  ``None`` switches it
    off; the environment variable ``QAMD_TREE_CACHE`` sets the initial directory."""
    _TREE_CACHE_DIR[0] = os.fspath(directory) if directory else None


#: bump when the candidate set / objective of a named strategy changes (r2: "auto-hq" gained recursive bisection;
#: r3: one per-step cost helper in the time model)
_FINDER_REV = 3


def _cache_file(inputs, output, size_dict, optimize):
    d = _TREE_CACHE_DIR[0]
    if d is None:
        d = os.environ.get("QAMD_TREE_CACHE") or None
    if not d:
        return None
    return os.path.join(d, f"tree-{geometry_hash(inputs, output, size_dict, optimize)}.json")


def find_path(inputs, output, size_dict, optimize="greedy", dtype="float32"):
    """Resolve quimb's ``optimize=`` argument to a ``ContractionTree``.  ``dtype`` matters to the time objective only
    (peaks, item size, which fused kernels exist): it prices the trees of ``"auto-time"`` and keys their cache entries."""
    if isinstance(optimize, ContractionTree):
        return optimize
    if hasattr(optimize, "get_path") and hasattr(optimize, "size_dict"):
        return ContractionTree.from_any(optimize, inputs, output, size_dict)
    if isinstance(optimize, str):
        if optimize not in ("greedy", "auto", "auto-hq", "random-greedy", "auto-time", "bisection"):
            raise ValueError(f"unknown contraction strategy {optimize!r}")
        import numpy as _np

        dt_name = _np.dtype(dtype).name
        # the finder's revision is part of the key: entries written by an older candidate set are searched again
        path_file = _cache_file(inputs, output, size_dict,
                                f"{optimize}#r{_FINDER_REV}" + ("@" + dt_name if optimize == "auto-time" else ""))
        if path_file and os.path.exists(path_file):
            try:
                with open(path_file) as f:
                    ssa = [tuple(c) for c in json.load(f)["ssa_path"]]
                return ContractionTree(inputs, output, size_dict, ssa_path=ssa)
            except (OSError, ValueError, KeyError, IndexError):
                pass                                        # unreadable / stale entry: search again
        if optimize in ("greedy", "auto"):
            tree = ContractionTree(inputs, output, size_dict, ssa_path=greedy_ssa(inputs, output, size_dict))
        elif optimize in ("auto-time", "auto-hq", "bisection"):
            # candidates: perturbed greedy restarts (good on irregular networks) and recursive bisection at a few
            # leaf sizes (good wherever the best tree is a join of compact regions: lattices); the objective picks
            timed = optimize == "auto-time"
            objective = (lambda t: modeled_time(t, dt_name)) if timed else (lambda t: t.contraction_cost())
            cands = []
            if optimize != "bisection":
                cands.append(random_greedy(inputs, output, size_dict, repeats=32 if timed else 64,
                                           minimize="time" if timed else "flops", dtype=dt_name))
            if len(inputs) > 8:
                for leaf in (16, 25, 32, 48):
                    if leaf < len(inputs) or not cands:
                        cands.append(ContractionTree(inputs, output, size_dict,
                                                     ssa_path=bisection_ssa(inputs, output, size_dict, leaf_size=leaf)))
            if not cands:
                cands.append(ContractionTree(inputs, output, size_dict, ssa_path=greedy_ssa(inputs, output, size_dict)))
            # (a wide tree can cost the time model more to plan than the others to run: rank by cost first)
            cands.sort(key=lambda t: t.contraction_cost())
            cands = [t for t in cands if t.contraction_cost() <= 64 * cands[0].contraction_cost()]
            tree = min(cands, key=objective)
        else:
            tree = random_greedy(inputs, output, size_dict, repeats=32)
        if path_file:
            os.makedirs(os.path.dirname(path_file), exist_ok=True)
            tmp = f"{path_file}.{os.getpid()}.tmp"
            with open(tmp, "w") as f:
                json.dump({"ssa_path": [list(c) for c in tree.ssa_path], "cost": tree.contraction_cost(),
                           "strategy": optimize}, f)
            os.replace(tmp, path_file)                      # atomic: concurrent ranks may race on the same key
        return tree
    if callable(optimize):
        path = optimize(inputs, output, size_dict)
        return ContractionTree(inputs, output, size_dict, path=path)
    # explicit linear path
    return ContractionTree(inputs, output, size_dict, path=list(optimize))


def find_slices(tree, target_slices=None, target_size=None, max_slices=1 << 20, minimize="flops", dtype="float32"):
    """Greedily pick indices to slice: at every round take the index whose
    removal gives the lowest total cost (per-slice cost x number of slices; ``minimize="time"``: the lowest
    ``modeled_time``, hoisted steps counted once), until ``target_slices`` slices / ``target_size`` max
    intermediate is reached."""
    if target_slices is None and target_size is None:
        raise ValueError("need target_slices or target_size")
    sliced = list(tree.sliced_inds)
    cur = tree
    out_set = set(tree.output)

    def done(t):
        ok = True
        if target_slices is not None:
            ok = ok and t.nslices >= target_slices
        if target_size is not None:
            ok = ok and t.max_size() <= target_size
        return ok

    while not done(cur) and cur.nslices < max_slices:
        cands = set()
        for con, res, ops, keep, _ in cur.steps:
            for t in ops:
                cands.update(t)
        cands -= out_set
        cands = [ix for ix in cands if tree.size_dict[ix] > 1]
        if not cands:
            break
        best = None
        for ix in sorted(cands, key=repr):
            t = tree.with_slices(sliced + [ix])
            key = ((t.contraction_cost() if minimize == "flops" else modeled_time(t, dtype)), t.max_size())
            if best is None or key < best[0]:
                best = (key, ix, t)
        sliced.append(best[1])
        cur = best[2]
    return cur


# ---------------------------------------------------------------------------------------------------------------
# recursive bisection: the finder for networks whose good trees are JOINS of compact regions (lattices)
# ---------------------------------------------------------------------------------------------------------------
def _nested_to_ssa(tree, n):
    """Post-order linearisation of a nested pairing of input ids into an ssa path."""
    ssa, nxt = [], [n]

    def walk(t):
        if not isinstance(t, tuple):
            return t
        ids = [walk(c) for c in t]
        cur = ids[0]
        for o in ids[1:]:
            ssa.append((cur, o))
            cur = nxt[0]
            nxt[0] += 1
        return cur

    walk(tree)
    return ssa


def _ssa_to_nested(ssa, ids):
    """The nested pairing an ssa path over the local tensors ``ids`` describes (local numbering 0..len-1)."""
    node = dict(enumerate(ids))
    nxt = len(ids)
    for con in ssa:
        node[nxt] = tuple(node.pop(c) for c in con)
        nxt += 1
    rest = list(node.values())
    return rest[0] if len(rest) == 1 else tuple(rest)


def _growth_sweep_ssa(inputs, output, size_dict, start):
    """One tensor grows by absorbing, at every step, the neighbour that leaves the SMALLEST result (ties: the
    cheapest step, then the lowest id) -- a boundary sweep whose front follows the geometry of the network."""
    n = len(inputs)
    out_set = set(output)
    counts = {}
    for t in inputs:
        for ix in set(t):
            counts[ix] = counts.get(ix, 0) + 1
    where = {}
    for i, t in enumerate(inputs):
        for ix in t:
            where.setdefault(ix, set()).add(i)
    cur = list(dict.fromkeys(inputs[start]))
    seen = {ix: 1 for ix in cur}
    left = set(range(n)) - {start}
    ssa, cur_id, nxt = [], start, n
    while left:
        cand = set()
        for ix in cur:
            cand |= where[ix] & left
        if not cand:
            cand = {min(left)}                       # disconnected: outer product with the next tensor
        best = None
        for j in cand:
            tj = inputs[j]
            keep = [ix for ix in cur if ix in out_set or counts[ix] > seen.get(ix, 0) + (1 if ix in tj else 0)]
            keep += [ix for ix in dict.fromkeys(tj) if ix not in seen and (ix in out_set or counts[ix] > 1)]
            size = prod(size_dict[ix] for ix in keep)
            cost = prod(size_dict[ix] for ix in set(cur) | set(tj))
            key = (size, cost, j)
            if best is None or key < best[0]:
                best = (key, j, keep)
        _, j, keep = best
        for ix in set(inputs[j]):
            seen[ix] = seen.get(ix, 0) + 1
        cur = keep
        ssa.append((cur_id, j))
        cur_id, nxt = nxt, nxt + 1
        left.discard(j)
    return ssa


def _bisect(nodes, adj, rng, tries=6, balance=0.1):
    """Balanced two-way partition of ``nodes`` with a small cut: region growing from several seeds, each refined by
    Fiduccia-Mattheyses passes (single-node moves by gain under the balance constraint); the lightest cut wins."""
    nodes = list(nodes)
    nset = set(nodes)
    half = len(nodes) // 2
    lo, hi = int(len(nodes) * (0.5 - balance)), int(len(nodes) * (0.5 + balance)) + 1
    lo = max(lo, 1)

    def cut_of(side):
        return sum(w for u in side for v, w in adj[u].items() if v in nset and v not in side)

    def grow(seed):
        side = {seed}
        frontier = {}
        for v, w in adj[seed].items():
            if v in nset:
                frontier[v] = frontier.get(v, 0.0) + w
        while len(side) < half:
            if frontier:
                v = max(frontier, key=lambda x: (frontier[x], -x))
                del frontier[v]
            else:
                v = next(x for x in nodes if x not in side)
            side.add(v)
            for u, w in adj[v].items():
                if u in nset and u not in side:
                    frontier[u] = frontier.get(u, 0.0) + w
        return side

    def refine(side):
        side = set(side)
        for _ in range(8):
            improved = False
            gains = []
            for u in nodes:
                inside = u in side
                ext = sum(w for v, w in adj[u].items() if v in nset and ((v in side) != inside))
                itn = sum(w for v, w in adj[u].items() if v in nset and ((v in side) == inside))
                gains.append((ext - itn, u))
            gains.sort(reverse=True)
            for g, u in gains:
                if g <= 1e-12:
                    break
                new_len = len(side) + (-1 if u in side else 1)
                if not (lo <= new_len <= hi):
                    continue
                # re-evaluate the gain against the current sides (earlier moves of this pass may have changed it)
                inside = u in side
                ext = sum(w for v, w in adj[u].items() if v in nset and ((v in side) != inside))
                itn = sum(w for v, w in adj[u].items() if v in nset and ((v in side) == inside))
                if ext - itn > 1e-12:
                    (side.discard if inside else side.add)(u)
                    improved = True
            if not improved:
                break
        return side

    # seeds: the ends of a double BFS sweep (peripheral nodes), then random ones
    def far(s):
        dist, q = {s: 0}, [s]
        for u in q:
            for v in adj[u]:
                if v in nset and v not in dist:
                    dist[v] = dist[u] + 1
                    q.append(v)
        return max(dist, key=lambda x: (dist[x], -x))

    a = far(nodes[0])
    seeds = [a, far(a)] + [rng.choice(nodes) for _ in range(max(tries - 2, 0))]
    best = None
    for s in seeds:
        side = refine(grow(s))
        c = cut_of(side)
        if best is None or c < best[0] - 1e-12:
            best = (c, side)
    side = best[1]
    return [u for u in nodes if u in side], [u for u in nodes if u not in side]


def bisection_ssa(inputs, output, size_dict, leaf_size=32, seed=0, repeats=8):
    """Recursive bisection with reconfigured leaves: split the network into balanced halves along a light cut until a
    part has at most ``leaf_size`` tensors, order every leaf with the cheapest of greedy / perturbed greedy /
    boundary-growth sweeps from each of its tensors (the leaf's cut bonds are its output), join the parts bottom-up.
    On a 10 x 10 lattice this is four 5 x 5 corner sweeps, two joins and a closing product -- ``quadrant_path_2d``
    found instead of written down.  (The reference leaves this to cotengra's hyper-optimisers,
    quimb/tensor/tensor_core.py:9004-9014.)"""
    import math

    rng = random.Random(seed)
    inputs = [tuple(t) for t in inputs]
    n = len(inputs)
    out_set = set(output)
    adj = {i: {} for i in range(n)}
    where = {}
    for i, t in enumerate(inputs):
        for ix in set(t):
            where.setdefault(ix, []).append(i)
    for ix, ts in where.items():
        w = math.log2(max(size_dict[ix], 1))
        for a in range(len(ts)):
            for b in range(a + 1, len(ts)):
                adj[ts[a]][ts[b]] = adj[ts[a]].get(ts[b], 0.0) + w
                adj[ts[b]][ts[a]] = adj[ts[b]].get(ts[a], 0.0) + w

    def leaf(nodes):
        if len(nodes) == 1:
            return nodes[0]
        inside = set(nodes)
        sub = [inputs[i] for i in nodes]
        sub_out = tuple(ix for ix in dict.fromkeys(ix for t in sub for ix in t)
                        if ix in out_set or any(j not in inside for j in where[ix]))
        cands = [greedy_ssa(sub, sub_out, size_dict)]
        for r in range(repeats):
            cands.append(greedy_ssa(sub, sub_out, size_dict, temperature=0.3 * rng.random(), rng=rng,
                                    costmod=rng.choice([0.5, 1.0, 2.0])))
        for s in range(len(nodes)):
            cands.append(_growth_sweep_ssa(sub, sub_out, size_dict, s))
        best = min(cands, key=lambda p: ContractionTree(sub, sub_out, size_dict, ssa_path=p).contraction_cost())
        return _ssa_to_nested(best, nodes)

    def solve(nodes):
        if len(nodes) <= leaf_size:
            return leaf(nodes)
        a, b = _bisect(nodes, adj, rng)
        if not a or not b:
            return leaf(nodes)
        return (solve(a), solve(b))

    return _nested_to_ssa(solve(list(range(n))), n)
