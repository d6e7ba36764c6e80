#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of a short bench.py run -> gpurun_out/prof_<tag>/ ; extra args go to bench.py
TAG=$1; shift
export TMPDIR=/tmp
rm -rf gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o r --output-format csv -- python bench.py --no-cpu-baseline --no-pmc --no-companion --no-configs --no-roofline --steps 60 --warmup 10 --blocks 2 --min-block-s 0.1 "$@" > gpurun_out/prof_$TAG.log 2>&1
echo "rocprof rc=$?"
