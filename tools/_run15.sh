mkdir -p gpurun_out
{
python bench.py --workload rollout --no-cpu-baseline --no-pmc --no-roofline --no-companion --steps 10 --warmup 3 --blocks 3 --min-block-s 0.3 2>gpurun_out/r5_run15.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('graph:', d['value'], d['unit'], d['ms_per_step'])"
python bench.py --workload rollout --no-graphs --no-cpu-baseline --no-pmc --no-roofline --no-companion --steps 10 --warmup 3 --blocks 3 --min-block-s 0.3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('eager:', d['value'], d['unit'], d['ms_per_step'])"
} > gpurun_out/r5_run15.txt 2>&1
cat gpurun_out/r5_run15.txt; tail -5 gpurun_out/r5_run15.err
