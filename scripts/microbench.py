#!/usr/bin/env python
"""Kernel micro-benchmarks (HIP events): GETT at compute- and HBM-bound shapes,
tiled permute.  Usage: python scripts/microbench.py [--quick]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quimb_amd as qa
from quimb_amd.pairwise import plan_pair
from quimb_amd.ops import run_pair_step

dev = qa.default_device()

def timeit(fn, reps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

def rnd(shape, dtype="float32"):
    t = torch.rand(int(np.prod(shape)), device=dev.tdev, dtype=torch.float32) - 0.5
    if dtype == "float64": t = t.double()
    return qa.Array(dev, t, shape, dtype)

def bench_pair(name, ai, ash, bi, bsh, oi, fixed=True, dtype="float32", cfgs=(-1,)):
    a, b = rnd(ash, dtype), rnd(bsh, dtype)
    step = plan_pair(tuple(ai), tuple(ash), tuple(bi), tuple(bsh), tuple(oi), fixed)
    g = step.spec
    out = qa.Array.empty(step.out_shape, dtype, dev)
    isz = np.dtype(dtype).itemsize
    flops = 2 * g.mults
    nbytes = isz * g.B * (g.M * g.K + g.K * g.N + g.M * g.N)
    for cfg in cfgs:
        # cfg == "T<n>": force the tiled GETT kernel with tile config n
        if isinstance(cfg, str):
            dev.force_kernel, cfg = -1, int(cfg[1:])
        else:
            dev.force_kernel = 0
        dev.force_tile_cfg = cfg
        try:
            t = timeit(lambda: run_pair_step(step, a, b, out))
        except Exception as e:
            print(f"{name:34s} cfg={cfg} FAILED {e}"); continue
        cp = [v for k, v in dev._pairs.items() if k[0] == g and k[4] == cfg and k[7] == dev.force_kernel][-1]
        print(f"{name:34s} kern={cp.struct.kernel} vc={cp.struct.vec_c} cfg={cp.struct.tile_cfg} sk={cp.struct.split_k} va={cp.struct.vec_a} vb={cp.struct.vec_b} "
              f"akc={cp.struct.a_kcontig} cn={cp.struct.c_ncontig} BMNK=({g.B},{g.M},{g.N},{g.K}) "
              f"{t*1e3:9.3f} ms {flops/t/1e12:8.2f} TF {nbytes/t/1e9:8.1f} GB/s", flush=True)
    dev.force_tile_cfg = -1
    dev.force_kernel = 0

quick = "--quick" in sys.argv
print("== GETT fp32: square GEMMs (compute-bound)")
for n in ([2048, 4096] if quick else [1024, 2048, 4096, 8192]):
    bench_pair(f"gemm {n}^3 NN", "mk", (n, n), "kn", (n, n), "mn", cfgs=(0, 1, 5))
bench_pair("gemm 4096^3 TN (A k-major)", "km", (4096, 4096), "kn", (4096, 4096), "mn", cfgs=(0,))
bench_pair("gemm 4096^3 NT", "mk", (4096, 4096), "nk", (4096, 4096), "mn", cfgs=(0,))
print("== GETT fp64")
bench_pair("dgemm 4096^3 NN", "mk", (4096, 4096), "kn", (4096, 4096), "mn", dtype="float64", cfgs=(0, 1, 5))
print("== PEPS sweep steps, D=6 (HBM-bound): A[L,h,v,R] x S[h,x,v,y] -> C[L,y?,x?,R]")
for (L, R) in [(6**4, 6**5), (6**8, 6), (6**9, 1), (1, 6**9), (6**2, 6**7)]:
    bench_pair(f"sweep L=6^{round(np.log(L)/np.log(6))} R=6^{round(np.log(R)/np.log(6))}", "lhvr", (L, 6, 6, R), "hxvy", (6, 6, 6, 6), "lxyr", fixed=False, cfgs=(-1, "T1"))
print("== sweep steps in the executor's death-ordered layout: A[h,v,(a),m] x S[h,x,v,y] -> C[x,(a),m,y]")
bench_pair("sweepZ m=6^9", "hvm", (6, 6, 6**9), "hxvy", (6, 6, 6, 6), "xmy", cfgs=(-1, "T1"))
bench_pair("sweepZ a=6 m=6^8", "havm", (6, 6, 6, 6**8), "hxvy", (6, 6, 6, 6), "axmy", cfgs=(-1,))
bench_pair("sweepX m=6^9 (C=[x,y,m])", "hvm", (6, 6, 6**9), "hxvy", (6, 6, 6, 6), "xym", cfgs=(-1,))
bench_pair("row-end K=36 N=6", "hvm", (6, 6, 6**9), "hvy", (6, 6, 6), "my", cfgs=(-1, "T3"))
bench_pair("row-start K=6 N=36", "vm", (6, 6**9), "xvy", (6, 6, 6), "xmy", cfgs=(-1, "T1"))
print("== merged 2-site step K=N=216")
bench_pair("sweep2 L=6^4 R=6^4", "lkr", (6**4, 216, 6**4), "kn", (216, 216), "lnr", fixed=True, cfgs=(0, 1))
bench_pair("K=216 N=36 L=6^4 R=6^4", "lkr", (6**4, 216, 6**4), "kn", (216, 36), "lnr", fixed=True, cfgs=(-1, "T1"))
print("== gate on state: psi[2^a,2,2^b] x G[2,2]")
bench_pair("1q gate on 2^28 state", "lkr", (2**14, 2, 2**13), "kn", (2, 2), "lnr", cfgs=(-1, "T3"))
bench_pair("2q gate on 2^28 state", "lkr", (2**13, 4, 2**13), "kn", (4, 4), "lnr", cfgs=(-1, "T3"))
bench_pair("f64 sweep L=6^4 R=6^4", "lhvr", (6**4, 6, 6, 6**4), "hxvy", (6, 6, 6, 6), "lxyr", fixed=False, dtype="float64", cfgs=(-1, "T1"))
print("== permute fp32")
def bench_perm(name, shape, perm):
    x = rnd(shape)
    t = timeit(lambda: x.transpose(perm))
    nb = 2 * 4 * int(np.prod(shape))
    print(f"{name:34s} {t*1e3:9.3f} ms {nb/t/1e9:8.1f} GB/s", flush=True)
bench_perm("2D 16384x16384 transpose", (16384, 16384), (1, 0))
bench_perm("6^11 reverse", (6,) * 11, tuple(reversed(range(11))))
bench_perm("6^11 swap last two", (6,) * 11, tuple(range(9)) + (10, 9))
bench_perm("6^11 rotate", (6,) * 11, tuple(range(1, 11)) + (0,))
bench_perm("[L,36,R] -> [L,R,36]", (6**4, 36, 6**5), (0, 2, 1))
print("== DMRG2 effective-Hamiltonian matvec (BASELINE config #5): chi=512, MPO bond 5, d=2, fp64")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import checks
tensors, left, right = checks.dmrg_effective_ham(512, dtype="float64")
A = qa.TNLinearOperator(tensors, left, right, optimize="random-greedy")
x = rnd((512 * 2 * 2 * 512,), "float64")
t = timeit(lambda: A.matvec(x), reps=10, warm=3)
tree = A._expr(0).tree
fl = tree.total_flops("float64")
print(f"matvec {t*1e3:8.3f} ms   {fl/t/1e12:6.2f} TF fp64   tree flops {fl:.3e} (survey: 1.095e10)   steps: "
      + ", ".join(f"({i.M}x{i.N}x{i.K})" for i in A._expr(0).executor.info), flush=True)
# per-step kernels and HIP-event times of the matvec
dev.profile = []
A.matvec(x)
dev.tdev and __import__("torch").cuda.synchronize()
for spec, dt_, name, sk, e0, e1 in dev.profile:
    shp = f"M={getattr(spec, 'M', '?')} N={getattr(spec, 'N', '?')} K={getattr(spec, 'K', '?')}"
    print(f"   {e0.elapsed_time(e1)*1e3:8.1f} us  {name}  split_k={sk}  {shp}")
dev.profile = None
# DMRG local solve: device Lanczos on a symmetrised chi=512 effective Hamiltonian (BASELINE config #5)
import time as _time
(L_, li), (W1_, w1i), (W2_, w2i), (R_, ri) = tensors
L_ = (L_ + L_.transpose(2, 1, 0)) / 2; R_ = (R_ + R_.transpose(2, 1, 0)) / 2
W1_ = (W1_ + W1_.transpose(0, 1, 3, 2)) / 2; W2_ = (W2_ + W2_.transpose(0, 1, 3, 2)) / 2
As = qa.TNLinearOperator([(L_, li), (W1_, w1i), (W2_, w2i), (R_, ri)], left, right, optimize="random-greedy")
v0 = rnd((512 * 2 * 2 * 512,), "float64")
qa.eigh_lanczos(As, k=1, which="SA", v0=v0, ncv=4, tol=1e-3, maxiter=8)   # warm the plan caches
__import__("torch").cuda.synchronize()
_mv = As.matvec
_cnt = [0]
def _counting(x):
    _cnt[0] += 1
    return _mv(x)
As.matvec = _counting
for ncv, tol_ in ((4, 1e-3), (20, 1e-8)):
    _cnt[0] = 0
    t0 = _time.perf_counter()
    w_, _v = qa.eigh_lanczos(As, k=1, which="SA", v0=v0, ncv=ncv, tol=tol_, maxiter=200)
    __import__("torch").cuda.synchronize()
    dt_ = _time.perf_counter() - t0
    print(f"device Lanczos chi=512 fp64 (ncv={ncv}, tol={tol_:g}): {dt_*1e3:8.2f} ms  {_cnt[0]} matvecs  "
          f"{dt_/_cnt[0]*1e3:.3f} ms/iteration  E0={w_[0]:.8f}", flush=True)
As.matvec = _mv
Ag = qa.TNLinearOperator([(L_, li), (W1_, w1i), (W2_, w2i), (R_, ri)], left, right, optimize="random-greedy", graph=True)
qa.eigh_lanczos(Ag, k=1, which="SA", v0=v0, ncv=20, tol=1e-8, maxiter=4)
__import__("torch").cuda.synchronize()
t0 = _time.perf_counter()
w_, _v = qa.eigh_lanczos(Ag, k=1, which="SA", v0=v0, ncv=20, tol=1e-8, maxiter=200)
__import__("torch").cuda.synchronize()
print(f"device Lanczos chi=512 fp64 (ncv=20, tol=1e-08, hipGraph matvec): {(_time.perf_counter()-t0)*1e3:8.2f} ms  E0={w_[0]:.8f}", flush=True)
dev.profile = []
qa.eigh_lanczos(As, k=1, which="SA", v0=v0, ncv=20, tol=1e-8, maxiter=3)
__import__("torch").cuda.synchronize()
agg = {}
for spec, dt_, name, sk, e0, e1 in dev.profile:
    key = (name, getattr(spec, "M", 0), getattr(spec, "N", 0), getattr(spec, "K", 0), sk)
    a_ = agg.setdefault(key, [0, 0.0]); a_[0] += 1; a_[1] += e0.elapsed_time(e1)
for key, (c_, t_) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"   lanczos pair kernels: {t_/c_*1e3:8.1f} us x{c_:3d}  {key}")
dev.profile = None
