"""How long do the rocSOLVER decompositions behind quimb_amd.linalg take at DMRG chi=512 sizes?"""
import time, torch
dev = torch.device("cuda")
x = torch.randn(1024, 1024, dtype=torch.float64, device=dev)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print(f"svd default      {t(lambda: torch.linalg.svd(x, full_matrices=False)):8.2f} ms")
for drv in ("gesvdj", "gesvda", "gesvd"):
    try:
        print(f"svd {drv:8s}     {t(lambda: torch.linalg.svd(x, full_matrices=False, driver=drv)):8.2f} ms")
    except Exception as e:
        print(f"svd {drv:8s}     unsupported ({type(e).__name__})")
g = x.T @ x
print(f"eigh 1024        {t(lambda: torch.linalg.eigh(g)):8.2f} ms")
print(f"qr 1024          {t(lambda: torch.linalg.qr(x)):8.2f} ms")
def svd_via_eigh():
    g = x.T @ x
    w, v = torch.linalg.eigh(g)
    s = w.clamp_min(0).sqrt().flip(0); v = v.flip(1)
    u = (x @ v) / s
    return u, s, v.T
print(f"svd via eigh(x^T x) {t(svd_via_eigh):8.2f} ms")
u, s, vh = svd_via_eigh(); s0 = torch.linalg.svdvals(x)
print("singular values rel err (largest / smallest):", float(abs(s[0]-s0[0])/s0[0]), float(abs(s[-1]-s0[-1])/s0[-1]))
xc = x.cpu()
t0 = time.perf_counter(); torch.linalg.svd(xc, full_matrices=False); print(f"svd on the host (LAPACK) {(time.perf_counter()-t0)*1e3:8.2f} ms")
