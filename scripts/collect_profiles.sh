#!/bin/bash
# Round-end evidence: rocprofv3 passes of the default bench command, summarised into
# gpurun_out/profiles_new/ (copy what should be judged into profiles/).
#   pass 1: --kernel-trace --stats         -> r01_bench_stats.txt (per-kernel durations)
#   pass 2: --pmc FETCH_SIZE               -> r01_bench_fetch.txt
#   pass 3: --pmc WRITE_SIZE               -> r01_bench_write.txt
#   pass 5: --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_32B_sum -> r01_bench_rdreq.txt
#           (read requests by size: the exact read byte count whatever the access width; FETCH_SIZE
#            alone tallies 128-B requests at 64 B)
#   calibration: the same three counters + FETCH_SIZE on scripts/probes/loadpat.hip, whose byte
#           count is known exactly -> r01_fetch_calibration.txt
#   pass 4: --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -> r01_bench_mfma.txt
# PMC passes never share a run with each other or with API traces (gpurun / MI355X_MICROARCH.md rules).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r01}
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-secondary"
pass() {  # name, rocprof args...
  n=$1; shift
  rm -rf /tmp/qprof_$n
  rocprofv3 "$@" -d /tmp/qprof_$n -o r -- $CMD > $OUT/${TAG}_bench_$n.json 2> /tmp/qprof_$n.log
  db=$(find /tmp/qprof_$n -name "r_results.db" | head -1)
  python $R/scripts/rocpd_summary.py $db --top 200 --by-grid > $OUT/${TAG}_bench_$n.txt 2>&1
  sed -i "s#/tmp/qprof_$n#rocprofv3 $* -- bench.py --steps 5 --warmup 2 --no-cpu --no-secondary#" $OUT/${TAG}_bench_$n.txt
}
pass stats --kernel-trace --stats
# the same step replayed as a launch program (one host call): under the profiler the Python launch loop falls behind the
# device, so the later corners' launches land BESIDE the first join and stretch it; replayed, the step has the shape of an
# unprofiled run (every corner queued before the first join) and both join launches agree with the bench's HIP events
CMD_SAVE="$CMD"; CMD="$CMD --launch program"
pass stats_program --kernel-trace --stats
CMD="$CMD_SAVE"
pass fetch --kernel-trace --pmc FETCH_SIZE
pass write --kernel-trace --pmc WRITE_SIZE
pass mfma  --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass rdreq --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_32B_sum
if hipcc --offload-arch=gfx950 -O3 $R/scripts/probes/loadpat.hip -o /tmp/loadpat 2>/dev/null; then
  for c in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_32B_sum"; do
    rm -rf /tmp/qcal; rocprofv3 --kernel-trace --pmc $c -d /tmp/qcal -o r -- /tmp/loadpat > /tmp/qcal.out 2>&1
    echo "## rocprofv3 --kernel-trace --pmc $c -- loadpat   (every probe<V,KS,DEPTH> launch reads 4*KS rows x M floats: 1451188224 bytes)"
    python $R/scripts/rocpd_summary.py $(find /tmp/qcal -name r_results.db | head -1) --top 60 | grep -v "^# /tmp"
  done > $OUT/${TAG}_fetch_calibration.txt
fi
python $R/scripts/make_traffic.py $OUT $TAG
ls -la $OUT
