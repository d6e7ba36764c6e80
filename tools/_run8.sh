mkdir -p gpurun_out
{
echo "== fwd kernels"; python tools/kbench.py --dtype bf16 --only fwd --layer conv_2d_1_2 2>&1 | grep -v amdgpu.ids | tail -4
echo "== default dgrad"; python tools/dgrad_bench.py --only L2 2>&1 | grep -v amdgpu.ids
echo "== gather"; DLWPCS_EDGE_COST=16 python tools/dgrad_bench.py --only L2 --gather 1 2>&1 | grep -v amdgpu.ids
echo "== gather e_any=0"; DLWPCS_EDGE_COST=16 DLWPCS_TUNE_OR=1048576 python tools/dgrad_bench.py --only L2 --gather 1 2>&1 | grep -v amdgpu.ids
echo "== gather e_two=0"; DLWPCS_EDGE_COST=16 DLWPCS_TUNE_OR=2097152 python tools/dgrad_bench.py --only L2 --gather 1 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r5_run8.txt 2>&1
cat gpurun_out/r5_run8.txt
