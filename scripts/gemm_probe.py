import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import quimb_amd as qa
from quimb_amd.pairwise import plan_pair
from quimb_amd.ops import run_pair_step
dev = qa.default_device()
n = 4096
dev.force_tile_cfg = int(os.environ.get("CFG", "1"))
a = qa.Array(dev, torch.rand(n*n, device=dev.tdev) - 0.5, (n, n), "float32")
b = qa.Array(dev, torch.rand(n*n, device=dev.tdev) - 0.5, (n, n), "float32")
step = plan_pair(("m","k"), (n,n), ("k","n"), (n,n), ("m","n"), True)
out = qa.Array.empty((n,n), "float32", dev)
for _ in range(5): run_pair_step(step, a, b, out)
torch.cuda.synchronize()
