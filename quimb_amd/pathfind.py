"""Contraction-order search (host, pure python).

In the reference this is cotengra's job (``optimize="greedy"`` is quimb's
default strategy, quimb/tensor/contraction.py:11; structured 2D networks use
explicit boundary sweeps, quimb/tensor/tn2d/core.py:1355-1484).  cotengra is not
part of this build, so the executor ships its own finders:

* ``greedy_path``      -- size-difference greedy (the classic opt_einsum rule)
* ``random_greedy``    -- Boltzmann-perturbed greedy restarts, keep the cheapest
* ``sweep_path_2d``    -- row-by-row, site-by-site boundary sweep of an Lx x Ly grid
* ``quadrant_path_2d`` -- four corner sweeps + two GEMM-shaped joins + a dot product (MFMA-bound, shardable)
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


# gfx950 roofs the time model prices a step against (MI355X_MICROARCH.md: dense MFMA peaks, achievable HBM copy rate)
_PEAK_FLOPS = {"float32": 157.3e12, "float64": 78.6e12, "complex64": 157.3e12, "complex128": 78.6e12}
_HBM_BYTES_PER_S = 6.29e12
_LAUNCH_S = 4e-6


def modeled_time(tree, dtype="float32"):
    """Roofline estimate of executing ``tree`` on one MI355X: every launch of the plan the whole-tree executor
    would issue costs ``max(flops / MFMA peak, algorithmic bytes / HBM rate) + launch``, slices repeated -- the
    objective of the intensity-aware finders.  Fused pairs (two big-x-small steps in
    one pass, chain2q.hip) count with the bytes they actually move, so an order that lines such pairs up is
    cheaper than one with the same multiplication count that does not."""
    import numpy as np

    from .executor import TreeExecutor   # planning only: no device is touched

    ex = TreeExecutor(tree, dtype)
    name = np.dtype(dtype).name
    f = 8 if np.dtype(dtype).kind == "c" else 2
    ns = tree.nslices
    t = 0.0
    for inf in ex.info:
        rep = ns if inf.sliced_dep else 1
        t += rep * (max(f * inf.mults / _PEAK_FLOPS[name], inf.bytes / _HBM_BYTES_PER_S) + _LAUNCH_S)
    return t


def fused_pair_count(tree, dtype="float32"):
    """Launches of the executor's plan that are fused pairs / triples of streaming steps."""
    from .executor import TreeExecutor

    return sum(1 for e in TreeExecutor(tree, dtype).plan if e[0] in ("chain2", "chain3"))


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
        if optimize not in ("greedy", "auto", "auto-hq", "random-greedy", "auto-time"):
            raise ValueError(f"unknown contraction strategy {optimize!r}")
        import numpy as _np

        dt_name = _np.dtype(dtype).name
        path_file = _cache_file(inputs, output, size_dict, optimize + ("@" + dt_name if optimize == "auto-time" else ""))
        if path_file and os.path.exists(path_file):
            try:
                with open(path_file) as f:
                    ssa = [tuple(c) for c in json.load(f)["ssa_path"]]
                return ContractionTree(inputs, output, size_dict, ssa_path=ssa)
            except (OSError, ValueError, KeyError, IndexError):
                pass                                        # unreadable / stale entry: search again
        if optimize in ("greedy", "auto"):
            tree = ContractionTree(inputs, output, size_dict, ssa_path=greedy_ssa(inputs, output, size_dict))
        elif optimize == "auto-time":
            tree = random_greedy(inputs, output, size_dict, repeats=32, minimize="time", dtype=dt_name)
        else:
            tree = random_greedy(inputs, output, size_dict, repeats=64 if optimize == "auto-hq" else 32)
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
