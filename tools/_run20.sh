mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_dgrad_gather.py -q 2>&1 | grep -v amdgpu.ids | tail -8
python tools/dgrad_bench.py --only L5,L6 2>&1 | grep -v amdgpu.ids
python tools/dgrad_bench.py --only L2,L5,L6,L9 --gather 1 2>&1 | grep -v amdgpu.ids
tools/ab.sh DLWPCS_OPTIONS=dgrad_gather=0 -- DLWPCS_X=0 -- DLWPCS_OPTIONS=dgrad_gather=0 -- DLWPCS_X=0
} > gpurun_out/r5_run20.txt 2>&1
cat gpurun_out/r5_run20.txt
