#!/bin/bash
# ONE guarded GPU call for a round's closing validation: every stage under its own `timeout`, everything logged under
# gpurun_out/ (merged back by gpurun), nothing that can rebuild the library or grow without bound.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_validate.sh r04'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-rNN}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== gpu tests" > $OUT/validate.log
timeout 700 python -u -m pytest tests -m gpu -q -x --timeout 150 --timeout-method thread > $OUT/gputest_$TAG.log 2>&1
echo "gpu tests rc=$? : $(tail -1 $OUT/gputest_$TAG.log)" >> $OUT/validate.log
echo "== smoke" >> $OUT/validate.log
timeout 90 python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/validate.log 2>&1
echo "== bench (default command)" >> $OUT/validate.log
timeout 400 python bench.py --steps 10 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench rc=$? : $(cut -c1-220 $OUT/bench_$TAG.json)" >> $OUT/validate.log
echo "== profiles" >> $OUT/validate.log
timeout 500 bash scripts/collect_profiles.sh $TAG > $OUT/collect_$TAG.log 2>&1
echo "profiles rc=$?" >> $OUT/validate.log
cat $OUT/validate.log
