#!/bin/bash
# PMC passes of `python bench.py` (eager, 3 steps) on the GPU box: one rocprofv3 run per counter group.
# usage: tools/gpu_pmc.sh <outdir-under-gpurun_out> <bench args...>     (counters via $PMC_GROUPS, ';'-separated)
set -u
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
GROUPS_="${PMC_GROUPS:-FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES}"
IFS=';' read -ra GS <<< "$GROUPS_"
i=0
for g in "${GS[@]}"; do
  d="$OUT/pass$i"
  echo "$g" > "$OUT/pass$i.counters"
  timeout 600 rocprofv3 --kernel-trace --pmc $g -d "$d" -o p --output-format csv -- \
      python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-graphs "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($g): rc=$?"
  i=$((i+1))
done
