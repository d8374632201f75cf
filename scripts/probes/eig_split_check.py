import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import quimb_amd as qa
g = np.load("tests/golden/split.npz")
x = g["x"]
X = qa.asarray(x)
for mb in (-1, 11, 5):
    u, s, vh = qa.linalg.svd_via_eig(X, mb)
    u, s, vh = u.to_numpy(), s.to_numpy(), vh.to_numpy()
    k = len(s)
    sref = np.linalg.svd(x, compute_uv=False)[:k]
    U, S, VH = np.linalg.svd(x, full_matrices=False)
    best = (U[:, :k] * S[:k]) @ VH[:k]
    print("max_bond", mb, "k", k, "s err", np.abs(s - sref).max(), "rec err vs best rank-k", np.abs((u * s) @ vh - best).max(),
          "orth U", np.abs(u.T @ u - np.eye(k)).max(), "orth V", np.abs(vh @ vh.T - np.eye(k)).max())
v = qa.asarray(np.arange(12.0).reshape(3, 4))
print(v[:, ::-1].to_numpy(), v[:, ::-1][:, :2].to_numpy(), v[::-1, 1:3].to_numpy())
