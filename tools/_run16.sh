mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "cfg3_bf16_training" 2>&1 | grep "cfg3 bf16\|per tensor\|passed\|failed"
echo "== padded grid"
DLWPCS_DGRAD_GATHER=0 timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "cfg3_bf16_training" 2>&1 | grep "cfg3 bf16\|per tensor\|passed\|failed"
} > gpurun_out/r5_run16.txt 2>&1
cat gpurun_out/r5_run16.txt
