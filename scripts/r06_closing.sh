#!/bin/bash
# Round 6 closing evidence in ONE guarded GPU call (everything under its own timeout, logged under gpurun_out/r06/):
#   gpurun --timeout 2700 -- 'bash scripts/r06_closing.sh'
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 bash scripts/gpu_validate.sh r06 > $O/validate_stdout.log 2>&1
cp gpurun_out/validate.log $O/r06_validate.log 2>/dev/null
cp gpurun_out/gputest_r06.log $O/r06_gputest_full.log 2>/dev/null
cp gpurun_out/bench_r06.json $O/r06_bench_line.json 2>/dev/null
timeout 600 bash scripts/bench_flags.sh > $O/r06_bench_flags.txt 2>&1
timeout 120 python scripts/probes/lane_trace.py > $O/r06_lane_trace.txt 2>&1
F="--offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=10000000 -DQAMD_RQ_ABLATE -I quimb_amd/csrc scripts/probes/rowq_probe.hip"
{
  echo "# scripts/probes/rowq_probe.hip: the LAST row of a corner sweep (S = 6^4, A[S][v1..v5] -> C[h][S][d1..d5]) on rowq_kernel<true, true>"
  echo "# 1) MFMA-rate calibration, then the kernel with parts switched off (RowArgs.pad2_ bits) and with fewer items"
  hipcc $F -o /tmp/rowq_probe 2>/dev/null && timeout 120 /tmp/rowq_probe
  echo
  echo "# 2) the same kernel with s_memtime stamps per phase (-DQAMD_RQ_TIMING): ticks per workgroup, summed over its ~10 items"
  hipcc $F -DQAMD_RQ_TIMING -o /tmp/rowq_probe_t 2>/dev/null && timeout 120 /tmp/rowq_probe_t
  echo
  echo "# 3) through the Python boundary (scripts/probes/rowq_time.py): whole-size and range-sliced new legs"
  timeout 120 python scripts/probes/rowq_time.py 2>/dev/null
} > $O/r06_rowq_probe.txt 2>&1
{
  echo "# bench.py --steps 30 --warmup 3 --no-cpu --no-secondary with QAMD_ROW_KERNEL = ... (ms per step, TFLOP/s)"
  for rk in quad quad-queue quad-prio3 quad-prio8 tile; do
    echo "== $rk"
    QAMD_ROW_KERNEL=$rk timeout 150 python bench.py --steps 30 --warmup 3 --no-cpu --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value'],2))"
  done
} > $O/r06_rowq_variants.txt 2>&1
cat $O/r06_validate.log; tail -3 $O/r06_gputest_full.log; cat $O/r06_rowq_variants.txt
