import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quimb_amd as qa
dev = qa.default_device()
rng = np.random.default_rng(0)
for (m, n, k) in [(256, 256, 48), (256, 256, 64), (256, 256, 128), (256, 256, 72), (512, 512, 256)]:
    a = rng.uniform(-0.5, 0.5, (k, m)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, (k, n)).astype(np.float32)
    want = a.astype(np.float64).T @ b.astype(np.float64)
    for tile in (44, 33, 22, 42, 24):
        dev.force_kernel, dev.force_tile_cfg = -5, 16 * (tile // 10) + tile % 10
        dev._pairs.clear()
        got = qa.einsum("km,kn->mn", qa.asarray(a), qa.asarray(b)).to_numpy().astype(np.float64)
        err = np.abs(got - want)
        bad = err > 1e-4
        print(f"{m}x{n}x{k} tile {tile}: maxerr {err.max():.3e} bad {bad.sum()} / {bad.size}", end="")
        if bad.any():
            rows = np.where(bad.any(axis=1))[0]; cols = np.where(bad.any(axis=0))[0]
            print(f" rows {rows[:8]}..{rows[-1]} ({len(rows)}) cols {cols[:8]}..{cols[-1]} ({len(cols)})", end="")
            # which k-tiles are missing? solve: got - want = -sum over missing k rows
            r, c = np.argwhere(bad)[0]
            contrib = a[:, r].astype(np.float64) * b[:, c].astype(np.float64)
            d = want[r, c] - got[r, c]
            # try to explain d as a sum of 2-row groups
            pairs = contrib.reshape(-1, 2).sum(1)
            print(f" | first bad ({r},{c}) diff {d:.4f}; k-pair contributions {np.round(pairs[:12], 3)}", end="")
        print()
