#!/bin/bash
# Extra end-of-round evidence in ONE guarded GPU call: the randomised stress drivers, a kernel trace of the DMRG local
# update (the Lanczos step kernels), the N > 1 bench path over gloo with the ranks sharing the GPU.
#   gpurun --timeout 1500 -- 'bash scripts/final_checks.sh'
ulimit -c 0
mkdir -p gpurun_out/final
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 300 python scripts/probes/stress_pairs.py > gpurun_out/final/stress_pairs.txt 2>&1 ; echo "rc=$?" >> gpurun_out/final/stress_pairs.txt )
( timeout 300 python scripts/probes/stress_trees.py > gpurun_out/final/stress_trees.txt 2>&1 ; echo "rc=$?" >> gpurun_out/final/stress_trees.txt )
( timeout 200 python scripts/probes/stress_rows.py 2026 120 > gpurun_out/final/stress_rows.txt 2>&1 ; echo "rc=$?" >> gpurun_out/final/stress_rows.txt )
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_dmrg -o r -- python $GRAFT_REPO_ROOT/scripts/dmrg_step.py > $GRAFT_REPO_ROOT/gpurun_out/final/dmrg_step_profiled.txt 2>&1 ; python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $(find /tmp/prof_dmrg -name r_results.db | head -1) --top 400 > $GRAFT_REPO_ROOT/gpurun_out/final/dmrg_step_stats.txt 2>&1 )
for n in 2 4; do
  # no launcher: bench.py starts its own ranks (torch.distributed.run underneath); gloo lets them share the one GPU of this box
  QAMD_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus $n --steps 5 --warmup 2 > gpurun_out/final/bench_gloo_${n}ranks.json 2> gpurun_out/final/bench_gloo_${n}ranks.err
  echo "gloo $n rc=$?"; cut -c1-300 gpurun_out/final/bench_gloo_${n}ranks.json
done
tail -n 3 gpurun_out/final/stress_pairs.txt; tail -n 3 gpurun_out/final/stress_trees.txt
head -20 gpurun_out/final/dmrg_step_stats.txt
