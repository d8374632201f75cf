cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c2
timeout 300 python -u -m pytest tests -m gpu -q -x --timeout 200 -p no:cacheprovider -k "decomp or dmrg or aliased or split" > gpurun_out/c2/gputest.log 2>&1; echo "gputest rc=$?" > gpurun_out/c2/status.txt
timeout 200 python scripts/probes/decomp_gemm.py > gpurun_out/c2/decomp_gemm.txt 2>&1; echo "probe rc=$?" >> gpurun_out/c2/status.txt
timeout 300 python scripts/dmrg_sweep.py 100 512 3 eig R qr > gpurun_out/c2/sweep_eig_qr.txt 2>&1; echo "sweep1 rc=$?" >> gpurun_out/c2/status.txt
timeout 300 python scripts/dmrg_sweep.py 100 512 3 rand R cholesky > gpurun_out/c2/sweep_rand_chol.txt 2>&1; echo "sweep2 rc=$?" >> gpurun_out/c2/status.txt
timeout 300 python scripts/dmrg_sweep.py 100 512 3 eig R cholesky > gpurun_out/c2/sweep_eig_chol.txt 2>&1; echo "sweep3 rc=$?" >> gpurun_out/c2/status.txt
cat gpurun_out/c2/status.txt; tail -8 gpurun_out/c2/gputest.log; cat gpurun_out/c2/decomp_gemm.txt; tail -12 gpurun_out/c2/sweep_*.txt
