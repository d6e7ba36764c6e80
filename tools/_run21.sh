mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_wgrad_batch.py tests/test_gpu_premask.py -q 2>&1 | grep -v amdgpu.ids | tail -4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KSTAT_ROWS=6 tools/kstat.sh cur DLWPCS_X=0
KSTAT_ROWS=3 KSTAT_ARGS="--dtype f32" tools/kstat.sh f32 DLWPCS_X=0
} > gpurun_out/r5_run21.txt 2>&1
cat gpurun_out/r5_run21.txt
