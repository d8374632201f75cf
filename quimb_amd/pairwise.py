"""Pairwise contraction planner (host logic, no device access).

Lowers one step of a contraction tree -- ``einsum(a_inds, b_inds -> out_inds)``
-- onto the GETT kernel's bundle description: every index is classified as
batch / M (only in ``a``) / N (only in ``b``) / K (contracted), runs of indices
that are adjacent in memory in every operand holding them are fused into one
group, and per-operand element strides are recorded.  No operand is permuted:
the transposes the reference materialises (cotengra's ``transpose -> reshape ->
matmul`` lowering behind quimb/tensor/contraction.py:285; quimb's own
``do("tensordot")`` at quimb/tensor/tensor_core.py:3793) become addressing.

Plans are pure data and cached, mirroring the reference's ``lru_cache`` on
``calc_fuse_perm_and_shape`` (quimb/tensor/array_ops.py:95) and ``inds_to_eq``
(quimb/tensor/contraction.py:103).
"""

import functools
from dataclasses import dataclass

MAX_GROUPS = 8


def contig_strides(shape):
    s = [1] * len(shape)
    for i in range(len(shape) - 2, -1, -1):
        s[i] = s[i + 1] * shape[i + 1]
    return tuple(s)


def prod(xs):
    p = 1
    for x in xs:
        p *= int(x)
    return p


@dataclass(frozen=True)
class ViewSpec:
    """A strided view of a contiguous array: unique ``inds`` with ``shape`` and
    element ``strides`` (repeated indices merged by summing strides = taking the
    diagonal)."""

    inds: tuple
    shape: tuple
    strides: tuple


@dataclass(frozen=True)
class ReduceSpec:
    """Single-operand preprocessing: sum ``red`` indices out of a view, giving a
    new contiguous array over ``keep``."""

    keep_inds: tuple
    keep_shape: tuple
    keep_strides: tuple
    red_shape: tuple
    red_strides: tuple


@dataclass(frozen=True)
class PermuteSpec:
    """Materialise a view as a contiguous array with the given index order."""

    inds: tuple
    shape: tuple
    strides: tuple  # source strides in the new order


@dataclass(frozen=True)
class GettSpec:
    """Bundle description handed to ``qamd_contract_pair``.  Each bundle is a
    tuple of groups, outermost first; group = (dim, stride_in_A, stride_in_B,
    stride_in_C) with ``None`` where an operand does not hold the group."""

    b: tuple
    m: tuple
    n: tuple
    k: tuple

    @property
    def B(self):
        return prod(g[0] for g in self.b)

    @property
    def M(self):
        return prod(g[0] for g in self.m)

    @property
    def N(self):
        return prod(g[0] for g in self.n)

    @property
    def K(self):
        return prod(g[0] for g in self.k)

    @property
    def mults(self):
        """Scalar multiplications = cotengra's ``contraction_cost`` unit
        (tests/test_tensor/test_tensor_core.py:1199-1205 in the reference)."""
        return self.B * self.M * self.N * self.K


@dataclass(frozen=True)
class BinarySpec:
    """Pure elementwise step (all indices are batch): out = a * b."""

    shape: tuple
    sa: tuple
    sb: tuple


@dataclass(frozen=True)
class PairStep:
    swapped: bool  # True: kernel operand A is the caller's second operand
    pre: tuple  # per caller operand: tuple of ReduceSpec / PermuteSpec to apply in order
    kind: str  # 'gett' | 'binary'
    spec: object
    out_inds: tuple
    out_shape: tuple
    mults: int


def _view(inds, shape):
    """Merge repeated indices (diagonal) into a strided view."""
    st = contig_strides(shape)
    uinds, ushape, ustr = [], [], []
    pos = {}
    for ix, d, s in zip(inds, shape, st):
        if ix in pos:
            j = pos[ix]
            if ushape[j] != d:
                raise ValueError(f"index {ix!r} has inconsistent sizes {ushape[j]} and {d}")
            ustr[j] += s
        else:
            pos[ix] = len(uinds)
            uinds.append(ix)
            ushape.append(int(d))
            ustr.append(s)
    return ViewSpec(tuple(uinds), tuple(ushape), tuple(ustr))


def _is_contig_view(v):
    return v.strides == contig_strides(v.shape)


def _fuse(order, size, stride_maps):
    """Fuse adjacent indices of ``order`` when contiguous in every operand of
    ``stride_maps`` (list of dict ind->stride).  Returns list of groups
    (dim, [stride per operand])."""
    groups = []
    for ix in order:
        d = size[ix]
        strs = [sm[ix] for sm in stride_maps]
        if d == 1:
            continue
        if groups:
            gd, gs = groups[-1]
            if all(p == s * d for p, s in zip(gs, strs)):
                groups[-1] = (gd * d, strs)
                continue
        groups.append((d, strs))
    return groups


def _prepare_operand(inds, shape, other_inds, out_set):
    """Diagonal-merge, then sum out indices private to this operand that are not
    in the output.  Returns (pre_ops, inds, shape, strides)."""
    v = _view(inds, shape)
    pre = []
    red = [ix for ix in v.inds if ix not in other_inds and ix not in out_set]
    if red:
        keep = [i for i, ix in enumerate(v.inds) if ix not in red]
        rix = [i for i, ix in enumerate(v.inds) if ix in red]
        spec = ReduceSpec(
            tuple(v.inds[i] for i in keep),
            tuple(v.shape[i] for i in keep),
            tuple(v.strides[i] for i in keep),
            tuple(v.shape[i] for i in rix),
            tuple(v.strides[i] for i in rix),
        )
        pre.append(spec)
        v = ViewSpec(spec.keep_inds, spec.keep_shape, contig_strides(spec.keep_shape))
    return pre, v


@functools.lru_cache(maxsize=2**14)
def plan_pair(a_inds, a_shape, b_inds, b_shape, out_inds, out_fixed=True, death=None):
    """Plan ``einsum(a, b -> out)``.

    Parameters
    ----------
    a_inds, b_inds : tuple of hashable
    a_shape, b_shape : tuple of int
    out_inds : tuple of hashable
        Indices of the result.  If ``out_fixed`` their order is binding;
        otherwise the planner picks the memory order of the result (the tree
        executor owns intermediates, so their layout is free).
    death : tuple of (ind, step) pairs, optional
        When each surviving index is next contracted; used to place soon-to-die
        indices innermost among the small operand's free indices.
    """
    a_inds, b_inds, out_inds = tuple(a_inds), tuple(b_inds), tuple(out_inds)
    out_set = set(out_inds)
    if len(out_set) != len(out_inds):
        raise ValueError("repeated output indices are not supported")
    pre_a, va = _prepare_operand(a_inds, tuple(a_shape), set(b_inds), out_set)
    pre_b, vb = _prepare_operand(b_inds, tuple(b_shape), set(a_inds), out_set)
    size = {}
    for v in (va, vb):
        for ix, d in zip(v.inds, v.shape):
            if size.setdefault(ix, d) != d:
                raise ValueError(f"index {ix!r} has inconsistent sizes {size[ix]} and {d}")
    for ix in out_inds:
        if ix not in size:
            raise ValueError(f"output index {ix!r} not found in inputs")

    sa_set, sb_set = set(va.inds), set(vb.inds)
    batch = [ix for ix in va.inds if ix in sb_set and ix in out_set]
    kk = [ix for ix in va.inds if ix in sb_set and ix not in out_set]
    mm = [ix for ix in va.inds if ix not in sb_set]
    nn = [ix for ix in vb.inds if ix not in sa_set]

    # ---- pure elementwise -------------------------------------------------
    if not kk and not mm and not nn:
        oi = out_inds if out_fixed else tuple(va.inds)
        sa = dict(zip(va.inds, va.strides))
        sb = dict(zip(vb.inds, vb.strides))
        shp = tuple(size[ix] for ix in oi)
        spec = BinarySpec(shp, tuple(sa[ix] for ix in oi), tuple(sb[ix] for ix in oi))
        return PairStep(False, (tuple(pre_a), tuple(pre_b)), "binary", spec, oi, shp, prod(shp))

    # ---- kernel operand A = the one with the larger free size --------------
    swapped = prod(size[i] for i in nn) > prod(size[i] for i in mm)
    if swapped:
        va, vb = vb, va
        mm, nn = nn, mm
        batch = [ix for ix in va.inds if ix in set(vb.inds) and ix in out_set]
        kk = [ix for ix in va.inds if ix in set(vb.inds) and ix not in out_set]
    sa = dict(zip(va.inds, va.strides))
    sb = dict(zip(vb.inds, vb.strides))

    # ---- result memory order -----------------------------------------------
    if out_fixed:
        oi = out_inds
    else:
        # "death-ordered" layout: indices that are contracted sooner sit further
        # out (slower).  The consumer then finds its contracted indices outermost and
        # adjacent (one fused K group) and everything it keeps as one long contiguous
        # run -- the streaming kernel's ideal operand -- and because every operand is
        # already death-ordered the surviving M indices keep their relative order, so
        # they fuse on both the load and the store side.  Stable: ties keep the big
        # operand's order, then the small operand's -- except in the group that dies NEXT
        # (the consumer's contraction bundle, the outermost block): a new index joins it at
        # its OUTER end, so that the surviving indices behind it stay one contiguous run in
        # the operand and in the result (open legs of a corner sweep, all contracted by one
        # later join: [h_new, h_old..., spectators..., d_new] instead of a run split in two).
        dmap = dict(death) if death else {}
        big = 1 << 60
        kset = set(kk)
        cand = [ix for ix in va.inds if ix not in kset] + list(nn)
        first = min((dmap.get(ix, big) for ix in cand), default=big)
        nset = set(nn)
        oi = tuple(sorted(cand, key=lambda ix: (dmap.get(ix, big),
                                                0 if (dmap.get(ix, big) == first and first < big and ix in nset) else 1)))
    oshape = tuple(size[ix] for ix in oi)
    sc = dict(zip(oi, contig_strides(oshape)))

    # ---- bundles -------------------------------------------------------------
    # iteration order of each bundle follows the big streamed side: A for m/k/b,
    # C for n (so stores along n run through memory in order)
    m_order = sorted(mm, key=lambda ix: -sa[ix])
    k_order = sorted(kk, key=lambda ix: -sa[ix])
    if not any(sa[ix] == 1 for ix in kk) and any(sb[ix] == 1 for ix in kk):
        # A is contiguous along a free index, B along a contracted one: the order of the k groups
        # is free for A, so let B's stride-1 group be the innermost (vector loads along k for B)
        k_order = sorted(kk, key=lambda ix: -sb[ix])
    b_order = sorted(batch, key=lambda ix: -sa[ix])
    n_order = sorted(nn, key=lambda ix: -sc[ix])
    gm = _fuse(m_order, size, [sa, sc])
    gk = _fuse(k_order, size, [sa, sb])
    gb = _fuse(b_order, size, [sa, sb, sc])
    gn = _fuse(n_order, size, [sb, sc])

    pre = [list(pre_a), list(pre_b)] if not swapped else [list(pre_b), list(pre_a)]
    # pre[0] belongs to kernel operand A, pre[1] to kernel operand B
    if max(len(gm), len(gk), len(gb), len(gn)) > MAX_GROUPS:
        # canonicalise: permute A -> [b, m, k], B -> [b, k, n] so every bundle fuses
        if out_fixed:
            # the result order is the caller's: run m and batch in THAT order, or the permuted operands would
            # still be scrambled against C (e.g. a dense vector built site by site, indices reversed)
            m_order = sorted(mm, key=lambda ix: -sc[ix])
            b_order = sorted(batch, key=lambda ix: -sc[ix])
        new_a = tuple(b_order + m_order + k_order)
        new_b = tuple(b_order + k_order + n_order)
        if tuple(va.inds) != new_a or not _is_contig_view(va):
            shp = tuple(size[ix] for ix in new_a)
            pre[0].append(PermuteSpec(new_a, shp, tuple(sa[ix] for ix in new_a)))
            va = ViewSpec(new_a, shp, contig_strides(shp))
            sa = dict(zip(va.inds, va.strides))
        if tuple(vb.inds) != new_b or not _is_contig_view(vb):
            shp = tuple(size[ix] for ix in new_b)
            pre[1].append(PermuteSpec(new_b, shp, tuple(sb[ix] for ix in new_b)))
            vb = ViewSpec(new_b, shp, contig_strides(shp))
            sb = dict(zip(vb.inds, vb.strides))
        if not out_fixed:
            oi = tuple(b_order + m_order + n_order)
            oshape = tuple(size[ix] for ix in oi)
            sc = dict(zip(oi, contig_strides(oshape)))
            n_order = sorted(nn, key=lambda ix: -sc[ix])
        gm = _fuse(m_order, size, [sa, sc])
        gk = _fuse(k_order, size, [sa, sb])
        gb = _fuse(b_order, size, [sa, sb, sc])
        gn = _fuse(n_order, size, [sb, sc])
        if max(len(gm), len(gk), len(gb), len(gn)) > MAX_GROUPS:
            raise NotImplementedError(
                "result layout needs more than %d index groups per bundle; "
                "request a different output order" % MAX_GROUPS
            )

    spec = GettSpec(
        b=tuple((d, s[0], s[1], s[2]) for d, s in gb),
        m=tuple((d, s[0], None, s[1]) for d, s in gm),
        n=tuple((d, None, s[0], s[1]) for d, s in gn),
        k=tuple((d, s[0], s[1], None) for d, s in gk),
    )
    pre_caller = (tuple(pre[1]), tuple(pre[0])) if swapped else (tuple(pre[0]), tuple(pre[1]))
    return PairStep(swapped, pre_caller, "gett", spec, oi, oshape, spec.mults)


def tensordot_inds(a_ndim, b_ndim, axes):
    """Index labels for numpy-style ``tensordot(a, b, axes)``: returns
    (a_inds, b_inds, out_inds) with the numpy output order (free a, free b)."""
    if isinstance(axes, int):
        ax_a = list(range(a_ndim - axes, a_ndim))
        ax_b = list(range(axes))
    else:
        ax_a, ax_b = axes
        ax_a = [ax_a] if isinstance(ax_a, int) else list(ax_a)
        ax_b = [ax_b] if isinstance(ax_b, int) else list(ax_b)
    ax_a = [x % a_ndim for x in ax_a]
    ax_b = [x % b_ndim for x in ax_b]
    if len(ax_a) != len(ax_b):
        raise ValueError("tensordot: axes lengths differ")
    a_inds = [("a", i) for i in range(a_ndim)]
    b_inds = [("b", i) for i in range(b_ndim)]
    for x, y in zip(ax_a, ax_b):
        b_inds[y] = a_inds[x]
    out = [ix for i, ix in enumerate(a_inds) if i not in ax_a]
    out += [ix for i, ix in enumerate(b_inds) if i not in ax_b]
    return tuple(a_inds), tuple(b_inds), tuple(out)


def parse_einsum(eq, nops):
    """Parse a numpy-style explicit/implicit einsum equation for ``nops``
    operands into (inputs, output) tuples of characters."""
    eq = eq.replace(" ", "")
    if "..." in eq:
        raise NotImplementedError("ellipsis in einsum equations is not supported")
    if "->" in eq:
        lhs, rhs = eq.split("->")
    else:
        lhs = eq
        counts = {}
        for c in lhs.replace(",", ""):
            counts[c] = counts.get(c, 0) + 1
        rhs = "".join(sorted(c for c, n in counts.items() if n == 1))
    terms = lhs.split(",")
    if len(terms) != nops:
        raise ValueError(f"einsum: equation has {len(terms)} terms but {nops} operands given")
    return tuple(tuple(t) for t in terms), tuple(rhs)


# ---------------------------------------------------------------------------
# fused pair of streaming steps (qamd_contract_chain2)
# ---------------------------------------------------------------------------
def chain2_chunk(dtype_name, D):
    """m-chunk of the fused pair kernel; mirrors ``qamd_chain2_chunk`` (0 = unsupported)."""
    if dtype_name == "float32":
        return 16 if 2 <= D <= 7 else 0
    if dtype_name == "float64":
        return 16 if 2 <= D <= 6 else 0
    return 0


def chain2_variants_ok(dtype_name, D):
    """Row-start / row-end pair shapes: mirrors ``qamd_chain2r_supported`` (fp32, D <= 6, 16-m chunk; a caller that
    pins the LDS-tile kernel -- ``Options.chain2_kernel = "lds"`` -- plans without them: ``plan_chain2(variants=False)``)."""
    return dtype_name == "float32" and 2 <= D <= 6 and chain2_chunk(dtype_name, D) == 16


@dataclass(frozen=True)
class Chain2Spec:
    """Two consecutive big-x-small steps fused:  X = A.W1 ; C = X.W2  (see chain2.hip)."""

    D: int
    m: tuple          # groups (dim, stride_in_A, stride_in_C), outermost first
    sa_v: int         # A stride of the carried index v
    off_k1: tuple     # A element offset of every k1 row (D*D entries)
    off_co: tuple     # C element offset of every n2_out value (D entries)
    w1_pack: PermuteSpec   # W1 -> [k1..., x, y] contiguous
    w2_pack: PermuteSpec   # W2 -> [y, v, n2_out, n2_in] contiguous
    out_inds: tuple
    out_shape: tuple
    mults: int        # scalar multiplications of both steps
    a_size: int
    c_size: int
    K1: int = 0       # rows of step 1: D*D (interior site: k1 = (h, u)) or D (row start: no h yet)
    NO: int = 0       # values of n2_out: D (interior site) or 1 (row end: n2 = n2_in only)

    @property
    def k1_single(self):
        return self.K1 == self.D

    @property
    def no_n2out(self):
        return self.NO == 1

    @property
    def M(self):
        return prod(g[0] for g in self.m)


@dataclass(frozen=True)
class RowpassSpec:
    """Five consecutive site absorptions of a boundary sweep fused into one launch (see rowpass.hip)."""

    D: int
    sv: tuple         # boundary-tensor strides of the five up legs
    sd: tuple         # result strides of the five new down legs
    sh: int           # result stride of the row's new open leg
    s_groups: tuple   # spectators: groups (dim, stride_in_A, stride_in_C), outermost first; None = the first row (no boundary tensor)
    w_strides: tuple  # per site: strides of (up, left bond, down, right bond) in the site tensor
    out_inds: tuple
    out_shape: tuple
    mults: int        # scalar multiplications of the five steps
    a_size: int
    c_size: int
    ed: tuple = (0,) * 5   # extents of the new down legs (0 = D): shorter for the range-sliced cut bonds of a rank's share
    eh: int = 0            # extent of the row's new open leg (0 = D)
    kernel: int = 0        # qamd_rowpass_plan.kernel: 0 auto, 1 rowpass.hip (16x16x4 tiles), 2 rowq.hip (4x4x1 multi-block MFMA)

ROWPASS_SITES = 5
ROWPASS_MAX_OUT = 1 << 24      # rowpass.hip (kernel 1) only: beyond that a row is bandwidth, not latency, for ITS item rate
ROW_KERNELS = {"auto": 0, "tile": 1, "quad": 2, "quad-queue": 3,      # 3: rowq.hip with its per-stream item queue (opt-in)
               "quad-prio3": 4, "quad-prio8": 5}                       # 4 / 5: experiments with the wave priorities of rowq.hip


def rowpass_supported(dtype_name, D, nsites):
    """(mirror of qamd_rowpass_supported)"""
    return dtype_name == "float32" and D == 6 and nsites == ROWPASS_SITES


def plan_rowpass(la, site_layouts, lc, size, dtype_name, kernel=0):
    """Try to express FIVE consecutive site absorptions of a boundary sweep as ONE ``qamd_contract_rowpass``
    (csrc/rowpass.hip; quimb absorbs the row site by site, quimb/tensor/tn2d/core.py:1393-1402).  ``la``: layout of the
    boundary tensor the first site meets, ``site_layouts``: the five site tensors' layouts in absorption order, ``lc``:
    the layout already chosen for the LAST step's result (all index tuples of C-contiguous arrays).  Returns a
    ``RowpassSpec`` or None when the steps do not have the row structure

        site 0: up v1 -> (down d1, bond b1);   site c: (up v, bond b_c) -> (down d, bond b_c+1);   site 4: -> (d5, open h)

    with every leg of the site tensors of one size D, the up legs taken from the boundary tensor itself, the bonds
    consumed by the next site and everything else of the boundary tensor untouched (the spectators S)."""
    if len(site_layouts) != ROWPASS_SITES:
        return None
    if la is None:
        return _plan_rowfirst(site_layouts, tuple(lc), size, dtype_name)
    kernel = ROW_KERNELS.get(kernel, kernel)
    la, lc = tuple(la), tuple(lc)
    if len(set(la)) != len(la) or len(set(lc)) != len(lc):
        return None
    sa = dict(zip(la, contig_strides(tuple(size[i] for i in la))))
    sc = dict(zip(lc, contig_strides(tuple(size[i] for i in lc))))
    cur, bond, D = set(la), None, None
    sv, sd, ws, sh, ed, eh = [], [], [], None, [], None
    touched = set()
    for c, lw in enumerate(site_layouts):
        lw = tuple(lw)
        if len(set(lw)) != len(lw) or (bond is not None and bond not in lw):
            return None
        touched |= set(lw)
        sw = dict(zip(lw, contig_strides(tuple(size[i] for i in lw))))
        ups = [ix for ix in lw if ix in cur and ix != bond]
        new = [ix for ix in lw if ix not in cur]
        if len(ups) != 1 or len(new) != 2 or len(lw) != (3 if bond is None else 4) or ups[0] not in sa:
            return None
        up = ups[0]
        D = size[up] if D is None else D
        if c < ROWPASS_SITES - 1:
            nb = [ix for ix in new if ix in site_layouts[c + 1]]       # the bond the NEXT site carries
            if len(nb) != 1:
                return None
            nbond = nb[0]
            down = new[0] if new[1] == nbond else new[1]
            if nbond in sc or down not in sc:
                return None
        else:
            down, nbond = new                                          # (d5, h): two open legs, either naming works --
            if down not in sc or nbond not in sc:
                return None
            if sc[down] > sc[nbond]:                                   # -- the INNER one of the result is d5: rowq.hip copies
                down, nbond = nbond, down                              # out whole (d2..d5) runs when they are contiguous
            sh = sc[nbond]
            eh = size[nbond]
        # up legs and bonds of one size D; the NEW open legs may be shorter (range-sliced cut bonds: quadrants.py)
        if size[up] != D or (bond is not None and size[bond] != D) or size[down] > D or size[nbond] > D \
                or (c < ROWPASS_SITES - 1 and size[nbond] != D):
            return None
        ed.append(size[down])
        sv.append(sa[up])
        sd.append(sc[down])
        ws.append((sw[up], sw[bond] if bond is not None else 0, sw[down], sw[nbond]))
        cur = (cur - {up, bond}) | {down, nbond}
        bond = nbond
    if not rowpass_supported(dtype_name, D, ROWPASS_SITES) or cur != set(lc):
        return None
    spect = [ix for ix in la if ix not in touched]
    gs = _fuse(spect, size, [sa, sc])
    n_s = prod(size[i] for i in spect)
    c_size = prod(size[i] for i in lc)
    full = all(e == D for e in ed) and eh == D
    if kernel == 0:
        kernel = 2
    if len(gs) > 4 or n_s * D >= 2**31 or (kernel == 1 and (c_size > ROWPASS_MAX_OUT or not full)):
        return None
    a_size = prod(size[i] for i in la)
    if kernel in (2, 3, 4, 5) and (a_size >= 2**31 or c_size >= 2**40):
        return None
    # multiplications of the five steps as they would have run one by one (all legs D: D^7 for the first site, D^8 for each
    # other): site c multiplies (columns: d_1..d_{c-1}, v_{c+1}..v_5) x (K: bond, v_c) x (N: d_c, next bond / h)
    mults, cols = 0, D**4
    for c in range(ROWPASS_SITES):
        k = D if c == 0 else D * D
        n = ed[c] * (eh if c == ROWPASS_SITES - 1 else D)
        mults += cols * k * n
        if c < ROWPASS_SITES - 1:
            cols = cols // D * ed[c]
    mults *= n_s
    return RowpassSpec(D=D, sv=tuple(sv), sd=tuple(sd), sh=sh, s_groups=tuple((d, st[0], st[1]) for d, st in gs),
                       w_strides=tuple(ws), out_inds=lc, out_shape=tuple(size[i] for i in lc), mults=mults,
                       a_size=a_size, c_size=c_size, ed=tuple(0 if e == D else e for e in ed), eh=0 if eh == D else eh,
                       kernel=kernel)


def _plan_rowfirst(site_layouts, lc, size, dtype_name):
    """The FIRST row of a sweep (``plan_rowpass(None, ...)``): no boundary tensor, the five site tensors -- the lattice edge,
    no up legs -- multiplied along their bonds:  site 0: (d1, b1);  site c: (b_c, d, b_c+1);  site 4: (b4, d5, h)."""
    if len(set(lc)) != len(lc):
        return None
    sc = dict(zip(lc, contig_strides(tuple(size[i] for i in lc))))
    bond, D = None, None
    sd, ws, sh, seen = [], [], None, set()
    for c, lw in enumerate(site_layouts):
        lw = tuple(lw)
        if len(set(lw)) != len(lw) or len(lw) != (2 if bond is None else 3) or (bond is not None and bond not in lw) \
                or seen & (set(lw) - {bond}):
            return None
        seen |= set(lw)
        sw = dict(zip(lw, contig_strides(tuple(size[i] for i in lw))))
        new = [ix for ix in lw if ix != bond]
        D = size[new[0]] if D is None else D
        if any(size[ix] != D for ix in lw):
            return None
        if c < ROWPASS_SITES - 1:
            nb = [ix for ix in new if ix in site_layouts[c + 1]]
            if len(nb) != 1:
                return None
            nbond = nb[0]
            down = new[0] if new[1] == nbond else new[1]
            if nbond in sc or down not in sc:
                return None
        else:
            down, nbond = new
            if down not in sc or nbond not in sc:
                return None
            sh = sc[nbond]
        sd.append(sc[down])
        ws.append((0, sw[bond] if bond is not None else 0, sw[down], sw[nbond]))
        bond = nbond
    if not rowpass_supported(dtype_name, D, ROWPASS_SITES) or len(lc) != ROWPASS_SITES + 1:
        return None
    if set(lc) != {ix for lw in site_layouts for ix in lw if ix in sc}:
        return None
    mults = D**3 + D**4 + D**5 + D**6 * D     # the four pairwise steps it replaces
    return RowpassSpec(D=D, sv=(0,) * ROWPASS_SITES, sd=tuple(sd), sh=sh, s_groups=None, w_strides=tuple(ws), out_inds=lc,
                       out_shape=tuple(size[i] for i in lc), mults=mults, a_size=0, c_size=prod(size[i] for i in lc))


def plan_chain2(la, l1, lx, l2, lc, size, dtype_name, variants=True):
    """Try to fuse  X[lx] = A[la].W1[l1]  and  C[lc] = X.W2[l2]  (all layouts are index
    tuples of C-contiguous arrays; ``lc`` is the layout already chosen for C).  Returns a
    ``Chain2Spec`` or None if the pair does not have the fusable structure."""
    s1, sx, s2, sc_ = set(l1), set(lx), set(l2), set(lc)
    if len(set(la)) != len(la) or len(s1) != len(l1) or len(s2) != len(l2):
        return None
    k1 = [ix for ix in la if ix in s1 and ix not in sx]
    if any(ix in s1 and ix in sx for ix in la):      # batch index in step 1
        return None
    n1 = [ix for ix in l1 if ix not in k1]
    if any(ix not in sx for ix in n1) or any(ix not in sx for ix in la if ix not in k1):
        return None                                    # step 1 sums something away
    k2 = [ix for ix in lx if ix in s2 and ix not in sc_]
    if any(ix in s2 and ix in sc_ for ix in lx):      # batch index in step 2
        return None
    n1set = set(n1)
    k2n = [ix for ix in k2 if ix in n1set]
    k2m = [ix for ix in k2 if ix not in n1set]
    xs = [ix for ix in n1 if ix not in k2]
    n2 = [ix for ix in l2 if ix not in k2]
    if any(ix not in sc_ for ix in n2) or any(ix not in sc_ for ix in lx if ix not in k2):
        return None
    if not (len(k2m) == 1 and len(k2n) == 1 and len(xs) == 1 and len(n2) in (1, 2)):
        return None
    D = size[k2m[0]]
    K1 = prod(size[ix] for ix in k1)
    if any(size[ix] != D for ix in k2n + xs + n2) or not k1 or any(size[ix] != D for ix in k1) or len(k1) > 2:
        return None
    # row-start (k1 = one index) and row-end (n2 = one index) shapes exist in the register kernel only
    variant = len(k1) == 1 or len(n2) == 1
    if variant and (len(k1) == 1 and len(n2) == 1 or not variants or not chain2_variants_ok(dtype_name, D)):
        return None
    chunk = chain2_chunk(dtype_name, D)
    if not chunk:
        return None
    v, y, x = k2m[0], k2n[0], xs[0]
    mm = [ix for ix in la if ix not in k1 and ix != v]
    if not mm or set(lc) != set(mm) | {x} | set(n2):
        return None
    sa = dict(zip(la, contig_strides(tuple(size[i] for i in la))))
    sc = dict(zip(lc, contig_strides(tuple(size[i] for i in lc))))
    # C must end with [.., m_inner, x, n2_in]
    n2in = lc[-1]
    if n2in not in n2 or lc[-2] != x:
        return None
    n2out = ([ix for ix in n2 if ix != n2in] or [None])[0]
    NO = D if n2out is not None else 1
    gm = _fuse(mm, size, [sa, sc])
    if not gm or len(gm) > MAX_GROUPS:
        return None
    d_in, (sa_in, sc_in) = gm[-1]
    if sa_in != 1 or sc_in != D * D or d_in % chunk:
        return None
    # byte offsets inside a chunk must fit 32 bits
    kshape = [size[ix] for ix in k1]
    kstr = [sa[ix] for ix in k1]
    off_k1 = [0]
    for d, st in zip(kshape, kstr):
        off_k1 = [o + i * st for o in off_k1 for i in range(d)]
    if (max(off_k1) + (D - 1) * sa[v] + 64) * 8 >= 2**32:
        return None
    s1d = dict(zip(l1, contig_strides(tuple(size[i] for i in l1))))
    s2d = dict(zip(l2, contig_strides(tuple(size[i] for i in l2))))
    off_co = tuple(i * sc[n2out] for i in range(D)) if n2out is not None else (0,)
    if variant and (any(o % 4 for o in off_co) or any(st[1] % 4 for _, st in gm[:-1])):
        return None                                    # the register kernel stores 16-byte vectors
    o1 = tuple(k1) + (x, y)
    o2 = (y, v, n2out, n2in) if n2out is not None else (y, v, n2in)
    w1_pack = PermuteSpec(o1, tuple(size[i] for i in o1), tuple(s1d[i] for i in o1))
    w2_pack = PermuteSpec(o2, tuple(size[i] for i in o2), tuple(s2d[i] for i in o2))
    M = prod(size[i] for i in mm)
    mults = M * D * K1 * (D * D) + M * D * (D * D) * (NO * D)   # step 1: M*D columns x K1 x N1 ; step 2: M*D (x) columns x K2 x N2
    return Chain2Spec(
        D=D,
        m=tuple((d, st[0], st[1]) for d, st in gm),
        sa_v=sa[v],
        off_k1=tuple(off_k1),
        off_co=off_co,
        w1_pack=w1_pack,
        w2_pack=w2_pack,
        out_inds=tuple(lc),
        out_shape=tuple(size[i] for i in lc),
        mults=mults,
        a_size=prod(size[i] for i in la),
        c_size=prod(size[i] for i in lc),
        K1=K1,
        NO=NO,
    )
