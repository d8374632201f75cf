"""One GEMM-shaped k-outer contraction, a few launches: the target of the rocprofv3 passes in scripts/collect_gemm_profiles.sh.
    python scripts/gemm_one.py M N K [tile] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import quimb_amd as qa
from quimb_amd.pairwise import plan_pair
from quimb_amd.ops import run_pair_step

m, n, k = (int(x) for x in sys.argv[1:4])
tile = sys.argv[4] if len(sys.argv) > 4 else "0"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = qa.default_device()
if tile != "0":      # pin the workgroup tile: plan inputs kernel = -5, tile_cfg = 16 ta + tb (the library reads no environment)
    dev.force_kernel, dev.force_tile_cfg = -5, 16 * (int(tile) // 10) + int(tile) % 10
fill = os.environ.get("QAMD_GEMM_FILL", "rand")
ta = (torch.rand(k, m, device=dev.tdev, dtype=torch.float32) - 0.5) if fill == "rand" else torch.zeros(k, m, device=dev.tdev)
tb = (torch.rand(k, n, device=dev.tdev, dtype=torch.float32) - 0.5) if fill == "rand" else torch.zeros(k, n, device=dev.tdev)
a = qa.Array(dev, ta.reshape(-1), (k, m), "float32")
b = qa.Array(dev, tb.reshape(-1), (k, n), "float32")
out = qa.Array.empty((m, n), "float32", dev)
step = plan_pair(("k", "m"), (k, m), ("k", "n"), (k, n), ("m", "n"), True)
for _ in range(2):
    run_pair_step(step, a, b, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run_pair_step(step, a, b, out)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / iters * 1e-3
name = dev.describe_pair(dev.compile_pair(step.spec, np.dtype("float32")))
print(f"{m}x{n}x{k} {name} fill={fill}: {t*1e3:.3f} ms  {2*m*n*k/t/1e12:.1f} TFLOP/s")
