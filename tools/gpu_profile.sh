#!/bin/bash
# On the GPU box: the bench line (un-profiled, live PMC passes) and the rocprofv3 kernel traces of the same workloads.
# usage: tools/gpu_profile.sh <tag>     -> gpurun_out/bench_<tag>.json, pmc_<tag>.json, prof_<tag>[_encoder6|_rollout]/
set -u
TAG=${1:-r03}
export TMPDIR=/tmp
rm -f gpurun_out/pmc_$TAG.json
python bench.py --pmc-out gpurun_out/pmc_$TAG.json > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?"
Q="--no-cpu-baseline --no-pmc --no-configs --no-roofline --no-dp-form --steps 60 --warmup 10 --blocks 2 --min-block-s 0.1"
rm -rf gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_encoder6 gpurun_out/prof_${TAG}_rollout gpurun_out/prof_${TAG}_unet2x2
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o r --output-format csv -- \
    python bench.py $Q > gpurun_out/prof_$TAG.log 2>&1
echo "rocprof unet2 rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_encoder6 -o r --output-format csv -- \
    python bench.py $Q --no-companion --workload encoder6 --channels 7 --dtype f32 > gpurun_out/prof_${TAG}_encoder6.log 2>&1
echo "rocprof encoder6 rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_rollout -o r --output-format csv -- \
    python bench.py --no-cpu-baseline --no-pmc --no-configs --no-roofline --no-dp-form --no-companion --workload rollout --steps 6 --warmup 2 --blocks 2 --min-block-s 0.05 > gpurun_out/prof_${TAG}_rollout.log 2>&1
echo "rocprof rollout rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_unet2x2 -o r --output-format csv -- \
    python bench.py $Q --no-companion --workload unet2x2 > gpurun_out/prof_${TAG}_unet2x2.log 2>&1
echo "rocprof unet2x2 rc=$?"
