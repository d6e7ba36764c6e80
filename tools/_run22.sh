mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/gpu_profile.sh r05 > gpurun_out/r5_profile.log 2>&1
cat gpurun_out/r5_profile.log; tail -3 gpurun_out/bench_r05.err
