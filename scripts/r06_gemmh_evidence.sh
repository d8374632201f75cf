#!/bin/bash
# Round 6, the opt-in split-product joins (csrc/gemmh.hip): probe, checks, PMC passes and the profiled bench in ONE guarded GPU
# call, logged under gpurun_out/r06h/ (copy what should be judged into profiles/):
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I quimb_amd/csrc scripts/probes/gemmh_probe.hip -o scripts/probes/_build/gemmh_probe
#   gpurun --timeout 1200 -- 'bash scripts/r06_gemmh_evidence.sh'
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r06h
mkdir -p $O
export TMPDIR=/tmp
P=scripts/probes/_build/gemmh_probe
{
  echo "# scripts/probes/gemmh_probe.hip: split + product + DOT epilogue against fp64 on the host (M N K ta tb iters fill centre)"
  timeout 300 $P | tail -8
  for t in "4 4" "3 4" "3 3"; do timeout 120 $P 7776 7776 7776 $t 5 0; done
  timeout 120 $P 8192 8192 8192 4 4 5 0
  timeout 120 $P 8192 8192 8192 4 4 5 2
  timeout 100 $P 1944 3888 7776 2 4 5 0
  timeout 100 $P 3888 3888 7776 3 4 5 0
} > $O/r06_gemmh_probe.txt 2>&1
{
  echo "# centred (8th argument 1) against uncentred (0) operands: max-norm and MEAN SIGNED error against fp64"
  for c in 1 0; do
    timeout 120 $P 7776 7776 7776 4 4 3 0 $c
    timeout 100 $P 2048 2048 7776 4 4 3 1 $c
    timeout 100 $P 2048 2048 512 4 4 3 0 $c
  done
} > $O/r06_gemmh_bias.txt 2>&1
timeout 300 python scripts/probes/gemmh_check.py > $O/r06_gemmh_check.txt 2>&1
{
  C="$P 7776 7776 7776 4 4 3 0"
  echo "## mfma"; bash scripts/pmc_probe.sh hm "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" $C
  echo "## waves"; bash scripts/pmc_probe.sh hw "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" $C
  echo "## lds"; bash scripts/pmc_probe.sh hl "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" $C
} > $O/r06_gemmh_pmc.txt 2>&1
rm -rf $R/gpurun_out/pmc_hm $R/gpurun_out/pmc_hw $R/gpurun_out/pmc_hl
# the whole step with the split-product joins under rocprofv3 (as a launch program: the shape of an unprofiled run)
cd /tmp
rm -rf /tmp/qprof_h
QAMD_JOIN_ARITH=f16x3 rocprofv3 --kernel-trace --stats -d /tmp/qprof_h -o r -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-secondary --tree quadrant --launch program > $O/r06_bench_f16x3_stats.json 2> /tmp/qprof_h.log
python $R/scripts/rocpd_summary.py $(find /tmp/qprof_h -name "r_results.db" | head -1) --top 200 --by-grid > $O/r06_bench_f16x3_stats.txt 2>&1
sed -i "s#/tmp/qprof_h#QAMD_JOIN_ARITH=f16x3 rocprofv3 --kernel-trace --stats -- bench.py --steps 5 --warmup 2 --no-cpu --no-secondary --tree quadrant --launch program#" $O/r06_bench_f16x3_stats.txt
cd $R
tail -12 $O/r06_gemmh_probe.txt; cat $O/r06_gemmh_bias.txt; tail -3 $O/r06_gemmh_check.txt; grep "gemmh8_kernel<4, 4, false>" $O/r06_gemmh_pmc.txt | sed 's/ \+/ /g'; head -30 $O/r06_bench_f16x3_stats.txt
