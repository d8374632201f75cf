#!/bin/bash
# rocprofv3 evidence for the permute kernel ("achieved HBM GB/s on the transpose"): durations, then FETCH_SIZE and
# WRITE_SIZE in their own passes (never combined with each other or with API traces) -> gpurun_out/profiles_new/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/bench_permute.py"
pass() {
  n=$1; shift
  rm -rf /tmp/qperm_$n
  rocprofv3 "$@" -d /tmp/qperm_$n -o r -- $CMD > $OUT/${TAG}_permute_$n.log 2> /tmp/qperm_$n.err
  db=$(find /tmp/qperm_$n -name "r_results.db" | head -1)
  python $R/scripts/rocpd_summary.py $db --top 40 --by-grid > $OUT/${TAG}_permute_$n.txt 2>&1
  sed -i "s#/tmp/qperm_$n#rocprofv3 $* -- scripts/bench_permute.py#" $OUT/${TAG}_permute_$n.txt
}
pass stats --kernel-trace --stats
pass fetch --kernel-trace --pmc FETCH_SIZE
pass write --kernel-trace --pmc WRITE_SIZE
ls -la $OUT | grep permute
