import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import quimb_amd as qa, checks
for dt in ("float32", "float64", "complex64", "complex128"):
    for seed in (1, 2, 3, 4):
        try:
            checks.check_random_pairs(dt, ncases=400, seed=seed)
            print(dt, seed, "ok", flush=True)
        except AssertionError as e:
            print(dt, seed, "FAIL", str(e)[:300], flush=True)
