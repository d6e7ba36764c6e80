mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | grep -v "amdgpu.ids" | tail -15 > gpurun_out/r5_suite19.txt
cat gpurun_out/r5_suite19.txt
