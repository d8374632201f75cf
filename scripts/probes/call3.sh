cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c3
timeout 300 python -u -m pytest tests -m gpu -q -x --timeout 200 -p no:cacheprovider -k "decomp or dmrg" > gpurun_out/c3/gputest.log 2>&1; echo "gputest rc=$?" > gpurun_out/c3/status.txt
timeout 200 python scripts/probes/decomp_gemm.py > gpurun_out/c3/decomp_gemm.txt 2>&1; echo "probe rc=$?" >> gpurun_out/c3/status.txt
timeout 300 python scripts/dmrg_sweep.py 100 512 4 rand R cholesky 0 > gpurun_out/c3/sweep_rand0_chol.txt 2>&1; echo "sweep rc=$?" >> gpurun_out/c3/status.txt
cat gpurun_out/c3/status.txt; tail -n 6 gpurun_out/c3/gputest.log; cat gpurun_out/c3/decomp_gemm.txt; tail -n 11 gpurun_out/c3/sweep_rand0_chol.txt | cut -c1-300
