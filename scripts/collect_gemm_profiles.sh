#!/bin/bash
# rocprofv3 passes of ONE GEMM-shaped join (scripts/gemm_one.py): kernel-trace stats, MFMA busy + clock, wave-cycle
# breakdown, HBM traffic.  usage: collect_gemm_profiles.sh TAG M N K [tile]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/gemm_one.py $*"
F=$OUT/${TAG}.txt
echo "# $CMD" > $F
$CMD >> $F 2>&1
QAMD_GEMM_FILL=zeros $CMD >> $F 2>&1
pass() {  # name, rocprof args...
  n=$1; shift
  rm -rf /tmp/qprof_$n
  rocprofv3 "$@" -d /tmp/qprof_$n -o r -- $CMD > /tmp/qprof_$n.out 2> /tmp/qprof_$n.log
  db=$(find /tmp/qprof_$n -name "r_results.db" | head -1)
  echo "## rocprofv3 $* -- gemm_one.py $CMD_ARGS" >> $F
  grep TFLOP /tmp/qprof_$n.out >> $F
  python $R/scripts/rocpd_summary.py $db --top 12 | grep -v "^# /tmp" >> $F 2>&1
}
CMD_ARGS="$*"
pass stats --kernel-trace --stats
pass mfma  --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass waves --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
pass lds --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS
pass fetch --kernel-trace --pmc FETCH_SIZE
pass write --kernel-trace --pmc WRITE_SIZE
cat $F
