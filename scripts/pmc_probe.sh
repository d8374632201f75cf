# usage: bash scripts/pmc_probe.sh <tag> "<counters>" <cmd...>   (one rocprofv3 --pmc pass, per-kernel summary)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=$1; shift; c=$1; shift
cd $R
rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$n -o r -- "$@" > /dev/null 2>&1
python scripts/rocpd_summary.py $(find $R/gpurun_out/pmc_$n -name "r_results.db" | head -1) --top 4 2>&1 | grep -v "^#" | cut -c1-40,100-200
