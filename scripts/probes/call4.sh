cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c4
timeout 900 python -u -m pytest tests -m gpu -q --timeout 200 --timeout-method thread -p no:cacheprovider > gpurun_out/c4/gputest.log 2>&1; echo "gputest rc=$?" > gpurun_out/c4/status.txt
timeout 300 python scripts/dmrg_sweep.py 100 512 4 rand R cholesky 0 > gpurun_out/c4/sweep_rand0_chol.txt 2>&1; echo "sweep rc=$?" >> gpurun_out/c4/status.txt
timeout 400 python bench.py > gpurun_out/c4/bench.json 2> gpurun_out/c4/bench.err; echo "bench rc=$?" >> gpurun_out/c4/status.txt
cat gpurun_out/c4/status.txt; tail -n 14 gpurun_out/c4/gputest.log; tail -n 11 gpurun_out/c4/sweep_rand0_chol.txt | cut -c1-300; cut -c1-400 gpurun_out/c4/bench.json
