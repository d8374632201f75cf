cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c5
timeout 600 python -u -m pytest tests -m gpu -q -x --timeout 200 -p no:cacheprovider -k "stream or tree or sharded or quadrant or full_size or program" > gpurun_out/c5/gputest.log 2>&1; echo "gputest rc=$?" > gpurun_out/c5/status.txt
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --no-cpu --no-secondary --steps 20 --tree quadrant > gpurun_out/c5/bench_q$q.json 2> gpurun_out/c5/bench_q$q.err; echo "bench q=$q rc=$? $(python -c "import json;d=json.load(open('gpurun_out/c5/bench_q$q.json'));print(d['ms_per_step'], d['value'])")" >> gpurun_out/c5/status.txt
done
timeout 200 python bench.py --no-cpu --no-secondary --steps 20 --emulate-world 8 > gpurun_out/c5/bench_w8.json 2> gpurun_out/c5/bench_w8.err; echo "w8 rc=$? $(python -c "import json;d=json.load(open('gpurun_out/c5/bench_w8.json'));print(d['ms_per_step'])")" >> gpurun_out/c5/status.txt
timeout 200 python bench.py --no-cpu --no-secondary --steps 20 --emulate-world 4 > gpurun_out/c5/bench_w4.json 2> gpurun_out/c5/bench_w4.err; echo "w4 rc=$? $(python -c "import json;d=json.load(open('gpurun_out/c5/bench_w4.json'));print(d['ms_per_step'])")" >> gpurun_out/c5/status.txt
timeout 120 python scripts/probes/lane_trace.py > gpurun_out/c5/lane_trace.txt 2>&1
QAMD_BENCH_KERNELS=1 timeout 200 python bench.py --no-cpu --no-secondary --steps 10 --tree quadrant > gpurun_out/c5/bench_k.json 2> gpurun_out/c5/bench_k.err
cat gpurun_out/c5/status.txt; tail -n 5 gpurun_out/c5/gputest.log
