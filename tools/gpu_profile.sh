#!/bin/bash
# On the GPU box: the bench line (un-profiled, live PMC passes) and the rocprofv3 kernel trace of the same command.
# usage: tools/gpu_profile.sh <tag>     -> gpurun_out/bench_<tag>.json, pmc_<tag>.json, prof_<tag>/
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
rm -f gpurun_out/pmc_$TAG.json
python bench.py --pmc-out gpurun_out/pmc_$TAG.json > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o r --output-format csv -- \
    python bench.py --no-cpu-baseline --no-pmc --steps 60 --warmup 10 --blocks 2 --min-block-s 0.1 > gpurun_out/prof_$TAG.log 2>&1
echo "rocprof rc=$?"
