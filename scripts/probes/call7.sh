cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c7
for rep in 1 2; do for pb in 0 4194304 16777216 100000000; do
  QAMD_HOLD_PREFIX=$pb timeout 200 python bench.py --no-cpu --no-secondary --steps 30 --tree quadrant > gpurun_out/c7/b_$pb.$rep.json 2> gpurun_out/c7/b_$pb.$rep.err
  echo "prefix=$pb rep=$rep: $(python -c "import json;d=json.load(open('gpurun_out/c7/b_$pb.$rep.json'));print(round(d['ms_per_step'],3), round(d['value'],2), d['result'].get('rel_err_vs_fp64_oracle'))")" >> gpurun_out/c7/status.txt
done; done
cat gpurun_out/c7/status.txt
