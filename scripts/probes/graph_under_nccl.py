"""hipGraph capture of the slice loop while an RCCL process group (and its watchdog thread) is alive --
the situation of every rank of `bench.py --gpus N`, reproduced with world_size 1 on a one-GPU box."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import faulthandler
faulthandler.dump_traceback_later(45, exit=True)      # a hang prints where it sits and exits
import numpy as np, torch, torch.distributed as dist
print("init", flush=True)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda")
for _ in range(3):
    dist.all_reduce(t)          # communicator + watchdog are live from here on
torch.cuda.synchronize()
print("process group up", flush=True)
import quimb_amd as qa
from oracle import np_oracle as orc
Lx, Ly, D = 6, 8, 4
arrays, inputs = orc.tn2d_rand(Lx, Ly, D, seed=2, dtype="float32")
size = {ix: D for tt in inputs for ix in tt}
tree = qa.find_slices(qa.ContractionTree(inputs, (), size, path=qa.sweep_path_2d(Lx, Ly)), target_slices=16)
ex = qa.TreeExecutor(tree, "float32")
xs = [qa.asarray(a) for a in arrays]
with warnings.catch_warnings():
    warnings.simplefilter("error")          # a capture failure would only warn and fall back: make it fatal here
    for rep in range(3):
        m, e = ex(xs, strip_exponent=True)
        print("contraction", rep, "done", flush=True)
        w = torch.tensor([m.item()], device="cuda", dtype=torch.float64)
        dist.all_reduce(w)
        time.sleep(0.2)                     # let the watchdog poll between and during replays
want = orc.oracle_array_contract([a.astype(np.float64) for a in arrays], inputs, (), path=qa.sweep_path_2d(Lx, Ly))
got = m.item() * 10.0 ** e
print("nslices", tree.nslices, "graphs:", len(getattr(ex, "_slice_graphs", {})), "value", got, "oracle", float(want),
      "rel", abs(got / float(want) - 1))
assert len(getattr(ex, "_slice_graphs", {})) == 1 and abs(got / float(want) - 1) < 1e-4
dist.destroy_process_group()
print("ok")
