ulimit -c 0
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout 150 python bench.py --no-cpu --no-secondary --steps 5 --warmup 2 "$@" 2>/tmp/err.txt | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t else None
print(None if d is None else (round(d['ms_per_step'],3), round(d['value'],2), d['result'].get('rel_err_vs_fp64_oracle'), d['config'].get('launch','')[:30]))
" || tail -3 /tmp/err.txt; }
run --tree sweep
run --inflight 2
run --emulate-world 8
run --emulate-world 8 --graph
run --emulate-world 8 --launch python
run --emulate-world 2
run --launch program
run --Lx 6 --Ly 6
run --sliced --slices 8
run --two-sided
