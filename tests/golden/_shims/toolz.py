"""Pure-python stand-in for the handful of toolz functions quimb imports
(quimb/utils.py:9-44).  Only used by tests/golden/make_golden.py."""
import itertools
from functools import reduce


def last(seq):
    return list(seq)[-1]


def concat(seqs):
    return itertools.chain.from_iterable(seqs)


def concatv(*seqs):
    return concat(seqs)


def frequencies(seq):
    d = {}
    for x in seq:
        d[x] = d.get(x, 0) + 1
    return d


def partition_all(n, seq):
    it = iter(seq)
    while True:
        chunk = tuple(itertools.islice(it, n))
        if not chunk:
            return
        yield chunk


def partition(n, seq):
    it = iter(seq)
    while True:
        chunk = tuple(itertools.islice(it, n))
        if len(chunk) < n:
            return
        yield chunk


def partitionby(func, seq):
    return map(tuple, (g for _, g in itertools.groupby(seq, key=func)))


def merge_with(func, *dicts):
    if len(dicts) == 1 and not isinstance(dicts[0], dict):
        dicts = dicts[0]
    out = {}
    for d in dicts:
        for k, v in d.items():
            out.setdefault(k, []).append(v)
    return {k: func(v) for k, v in out.items()}


def valmap(func, d):
    return {k: func(v) for k, v in d.items()}


def keymap(func, d):
    return {func(k): v for k, v in d.items()}


def compose(*funcs):
    if not funcs:
        return identity
    def composed(*a, **k):
        out = funcs[-1](*a, **k)
        for f in reversed(funcs[:-1]):
            out = f(out)
        return out
    return composed


def identity(x):
    return x


def isiterable(x):
    try:
        iter(x)
        return True
    except TypeError:
        return False


def unique(seq, key=None):
    seen = set()
    for x in seq:
        k = x if key is None else key(x)
        if k not in seen:
            seen.add(k)
            yield x
