"""One fp64 GEMM-shaped contraction, a few launches: the target of rocprofv3 passes on gemmd.hip / gettf.hip.
    python scripts/gemmd_one.py EQ dims...   e.g.  "mk,kn->mn" m=4096 k=4096 n=4096 [gemmd=0|1] [iters=5]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kw = dict(a.split("=") for a in sys.argv[2:])
use_gemmd = kw.pop("gemmd", "1") != "0"
iters = int(kw.pop("iters", 5))
import numpy as np
import torch

import quimb_amd as qa

eq = sys.argv[1]
dims = {k: int(v) for k, v in kw.items()}
lhs, out = eq.split("->")
ai, bi = lhs.split(",")
rng = np.random.default_rng(0)
if not use_gemmd:
    qa.default_device().force_kernel = -2      # plan input: automatic choice without the MFMA GEMM kernels
a = qa.asarray(rng.uniform(-0.5, 1.0, [dims[c] for c in ai]))
b = qa.asarray(rng.uniform(-0.5, 1.0, [dims[c] for c in bi]))
for _ in range(iters):
    qa.einsum(eq, a, b)
torch.cuda.synchronize()
print("done")
