"""GEMM-shaped contractions: the tiled GETT kernel (every tile config) next to torch.matmul
(rocBLAS / hipBLASLt) on the same operands -- the library is the yardstick, not a code path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import quimb_amd as qa
from quimb_amd.pairwise import plan_pair
from quimb_amd.ops import run_pair_step

dev = qa.default_device()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (2048, 2560, 512), (1024, 1024, 8192)]
for dt in ("float32", "float64"):
    tdt = torch.float32 if dt == "float32" else torch.float64
    for (m, n, k) in shapes:
        if dt == "float64" and m == 8192:
            continue
        ta = torch.rand(m, k, device=dev.tdev, dtype=tdt) - 0.5
        tb = torch.rand(k, n, device=dev.tdev, dtype=tdt) - 0.5
        t_lib = timeit(lambda: torch.matmul(ta, tb))
        a = qa.Array(dev, ta.reshape(-1), (m, k), dt)
        b = qa.Array(dev, tb.reshape(-1), (k, n), dt)
        out = qa.Array.empty((m, n), dt, dev)
        step = plan_pair(("m", "k"), (m, k), ("k", "n"), (k, n), ("m", "n"), True)
        res = []
        for cfg in (1, 6, 7):
            dev.force_tile_cfg = cfg
            dev._pairs.clear()
            try:
                t = timeit(lambda: run_pair_step(step, a, b, out))
                res.append(f"cfg{cfg} {2*m*n*k/t/1e12:6.1f}")
            except Exception as e:
                res.append(f"cfg{cfg} fail")
        dev.force_tile_cfg = None
        err = (out._buf.reshape(m, n) - torch.matmul(ta, tb)).abs().max().item()
        print(f"{dt} {m}x{n}x{k}: library {2*m*n*k/t_lib/1e12:6.1f} TF | gett " + " ".join(res) + f" TF | maxdiff {err:.2e}")
