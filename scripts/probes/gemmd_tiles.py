import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import quimb_amd as qa
dev = qa.default_device()
rng = np.random.default_rng(0)
chi, w, d = 512, 5, 2
CASES = [("apA,Astb->apstb", dict(a=chi, p=w, A=chi, s=d, t=d, b=chi)), ("arstB,brB->astb", dict(a=chi, r=w, s=d, t=d, B=chi, b=chi))]
for eq, dims in CASES:
    lhs, out = eq.split("->"); ai, bi = lhs.split(",")
    a = qa.asarray(rng.uniform(-0.5, 1.0, [dims[c] for c in ai])); b = qa.asarray(rng.uniform(-0.5, 1.0, [dims[c] for c in bi]))
    flop = 2.0 * np.prod([dims[c] for c in set(ai) | set(bi)])
    for tile in ("21", "31", "41", "51", "22", "32", "42", "52"):
        for sk in ("0", "1", "2", "4", "8"):
            dev.force_kernel, dev.force_tile_cfg = -6, 16 * int(tile[0]) + int(tile[1])
            dev.force_split_k = int(sk)
            dev._pairs.clear()
            dev.profile = []
            try:
                qa.einsum(eq, a, b)
            except Exception as e:
                print(eq, tile, sk, "ERR", e); dev.profile = None; continue
            name = dev.profile[-1][2]; dev.profile = None
            for _ in range(3): qa.einsum(eq, a, b)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): qa.einsum(eq, a, b)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
            print(f"{eq:18s} tile {tile} split {sk}: {dt*1e6:7.1f} us {flop/dt/1e12:5.1f} TF  {name}")
