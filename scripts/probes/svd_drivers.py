"""torch.linalg.svd on ROCm: reconstruction error per driver for small, rank-deficient fp32 matrices
(the boundary-MPS tensors of an Ising partition function are exactly that)."""
import numpy as np
import torch

torch.manual_seed(0)
dev = torch.device("cuda")
for (m, n, rank) in ((4, 2, 1), (4, 4, 2), (4, 8, 3), (16, 32, 6), (64, 64, 64), (216, 1296, 100)):
    a = (torch.randn(m, rank, dtype=torch.float64) @ torch.randn(rank, n, dtype=torch.float64))
    for dt in (torch.float32, torch.float64):
        x = a.to(dt).to(dev)
        row = []
        for drv in (None, "gesvd", "gesvdj", "gesvda"):
            try:
                u, s, vh = torch.linalg.svd(x, full_matrices=False, driver=drv)
                err = ((u * s) @ vh - x).abs().max().item() / x.abs().max().item()
                sref = np.linalg.svd(a.numpy(), compute_uv=False)[: len(s)]
                serr = np.abs(s.cpu().double().numpy() - sref).max() / sref[0]
                row.append("%s: rec %.1e sv %.1e" % (drv, err, serr))
            except Exception as ex:  # noqa: BLE001
                row.append("%s: %s" % (drv, type(ex).__name__))
        print((m, n, rank), str(dt).split(".")[1], " | ".join(row), flush=True)
