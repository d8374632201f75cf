#!/bin/bash
# Run chunk(s) of the -m gpu suite, one pytest process per chunk, logs under gpurun_out/$OUT (debugging aid: a box that
# dies under one chunk names the culprit). usage: scripts/run_gpu_chunk.sh OUT N [N ...]
set -f
out=gpurun_out/$1; shift
mkdir -p $out
for n in "$@"; do
  timeout 900 python -m pytest $(cat scripts/chunks/chunk$n.txt) -q -x --timeout 600 --durations=8 -p no:cacheprovider -s > $out/chunk$n.log 2>&1
  echo "chunk $n rc=$?" | tee -a $out/summary.txt
  tail -3 $out/chunk$n.log
done
