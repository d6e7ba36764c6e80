#!/bin/bash
# The headline step at other batch sizes per GPU (same model, same kernels): how much of the batch-32 step is per-launch cost.
# usage (GPU box): bash tools/batch_sweep.sh > gpurun_out/batch_sweep.txt
Q="--no-companion --no-cpu-baseline --no-configs --no-roofline --no-dp-form --no-pmc --blocks 3"
for b in 8 16 32 64 128; do
  out=$(timeout 300 python bench.py $Q --batch $b 2>/dev/null | tail -1)
  echo "batch $b | $(echo "$out" | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print("%.4f ms/step  %.1f samples/s  %.1f TFLOP/s" % (d["ms_per_step"], d["value"], d["model_tflops"]))
except Exception as e:
    print("FAILED", e)')"
done
