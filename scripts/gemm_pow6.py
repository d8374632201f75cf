"""GEMM-shaped joins of contraction trees at the workload's own extents (powers of 6 and their shards):
gemmk.hip (every workgroup tile) next to the older tiled GETT kernels and to torch.matmul (rocBLAS / hipBLASLt:
the yardstick, not a code path).  Operands are "k-outer": A[k, m], B[k, n] -> C[m, n]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import quimb_amd as qa
from quimb_amd.pairwise import plan_pair
from quimb_amd.ops import run_pair_step

dev = qa.default_device()
quick = "--quick" in sys.argv


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


shapes = [(7776, 7776, 7776), (3888, 1944, 7776), (1944, 3888, 7776), (972, 7776, 7776), (3888, 7776, 7776), (3888, 3888, 7776),
          (1296, 1296, 7776), (7776, 7776, 1296), (4096, 4096, 4096), (8192, 8192, 8192), (2048, 2560, 512)]
if quick:
    shapes = shapes[:3] + shapes[7:8]
tiles = [None, 44, 43, 34, 33, 42, 24, 32, 22]
for (m, n, k) in shapes:
    ta = torch.rand(k, m, device=dev.tdev, dtype=torch.float32) - 0.5
    tb = torch.rand(k, n, device=dev.tdev, dtype=torch.float32) - 0.5
    t_lib = timeit(lambda: torch.matmul(ta.t(), tb))
    ref = torch.matmul(ta.t().double()[:256], tb.double())       # fp64 reference of the first 256 rows
    a = qa.Array(dev, ta.reshape(-1), (k, m), "float32")
    b = qa.Array(dev, tb.reshape(-1), (k, n), "float32")
    out = qa.Array.empty((m, n), "float32", dev)
    step = plan_pair(("k", "m"), (k, m), ("k", "n"), (k, n), ("m", "n"), True)
    res = []
    dev.force_kernel, dev.force_tile_cfg = -2, -1          # plan input: automatic choice WITHOUT the MFMA GEMM kernels
    dev._pairs.clear()
    t = timeit(lambda: run_pair_step(step, a, b, out))
    res.append(f"old {2*m*n*k/t/1e12:6.1f}")
    dev.force_kernel = 0
    errs = []
    for tl in tiles:
        dev.force_kernel, dev.force_tile_cfg = (0, -1) if tl is None else (-5, 16 * (int(tl) // 10) + int(tl) % 10)
        dev._pairs.clear()
        out._buf.zero_()
        try:
            t = timeit(lambda: run_pair_step(step, a, b, out))
            name = dev.describe_pair(dev.compile_pair(step.spec, np.dtype("float32")))
            err = (out._buf.reshape(m, n)[:256].double() - ref).abs().max().item()
            errs.append(err)
            res.append(f"{'auto:' + name.split('<')[1][:4] if tl is None else tl} {2*m*n*k/t/1e12:6.1f}")
        except Exception as e:
            res.append(f"{tl} fail({e})")
    dev.force_kernel, dev.force_tile_cfg = 0, -1
    print(f"f32 {m}x{n}x{k}: library {2*m*n*k/t_lib/1e12:6.1f} TF | " + " | ".join(res) + f" | maxerr {max(errs):.2e} (K*2e-8={k*2e-8:.1e})", flush=True)
