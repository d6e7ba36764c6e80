#!/bin/bash
# head fold: parity tests, rollout A/B, training-step A/B (the forward kernel's register count moved 164 -> 220)
cd /root/repo
mkdir -p gpurun_out
{
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_head_fold.py tests/test_gpu_premask.py -x -q 2>&1 | tail -15
RB="python bench.py --workload rollout --no-cpu-baseline --no-pmc --no-companion --no-configs --no-roofline --no-dp-form --blocks 3 --min-block-s 0.5"
for f in 1 0 1 0; do echo "== rollout fold_head=$f"; DLWPCS_OPTIONS=fold_head=$f $RB 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.4f ms  %.1f %s' % (d['ms_per_step'], d['value'], d['unit']))"; done
echo "== training step"; bash tools/ab.sh X=1 -- X=2 --
} > gpurun_out/run23.txt 2>&1
