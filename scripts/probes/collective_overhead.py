"""What the ONE collective of a sharded contraction costs a rank per step: a rank-of-8 share as a launch program, timed (a) alone
(`qrank(xs, defer=True)`, what `--emulate-world` times), (b) through `contract_quadrants` in an RCCL group of ONE rank (the same
all-gather call and host read a rank of `--gpus 8` makes; one GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
import quimb_amd as qa
from bench import build_network
from quimb_amd.quadrants import QuadrantRank, QuadrantSharding, contract_quadrants

dist.init_process_group("nccl", rank=0, world_size=1)
arrays, inputs, size = build_network(10, 10, 6, 7, "float32")
sh = QuadrantSharding(inputs, size, 10, 10, 8)
r = int(np.argmax(sh.cost_report()["per_rank_mults"]))
qr = QuadrantRank(sh, r, "float32")
xs = sh.shard([qa.asarray(a) for a in arrays], r)
qr.program(xs)
def timed(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
a = timed(lambda: qr(xs, defer=True))
b = timed(lambda: contract_quadrants(qr, xs, strip_exponent=True))
print(f"share alone: {a:.3f} ms per step;  + all-gather in a group of one rank + host read: {b:.3f} ms per step  (+{(b - a) * 1e3:.0f} us)")
dist.destroy_process_group()
