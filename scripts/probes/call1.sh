cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c1
timeout 900 python -u -m pytest tests -m gpu -q --timeout 200 --timeout-method thread -p no:cacheprovider > gpurun_out/c1/gputest.log 2>&1; echo "gputest rc=$?" > gpurun_out/c1/status.txt
timeout 60 scripts/probes/bin/gridbar > gpurun_out/c1/gridbar.txt 2>&1; echo "gridbar rc=$?" >> gpurun_out/c1/status.txt
timeout 120 python scripts/probes/lane_trace.py > gpurun_out/c1/lane_trace.txt 2>&1; echo "lane rc=$?" >> gpurun_out/c1/status.txt
timeout 400 python bench.py > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err; echo "bench rc=$?" >> gpurun_out/c1/status.txt
cat gpurun_out/c1/status.txt; tail -25 gpurun_out/c1/gputest.log
