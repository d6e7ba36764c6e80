#!/bin/bash
# per-kernel average durations of the bf16 training step under rocprofv3 (hipGraph replays), for A/B runs of tune bits
# usage: tools/kstat.sh <tag> [ENV=VAL ...]
TAG=$1; shift
export TMPDIR=/tmp
rm -rf gpurun_out/ks_$TAG
env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/ks_$TAG -o r --output-format csv -- python bench.py --no-cpu-baseline --no-pmc --no-configs --no-roofline --no-dp-form --no-companion --steps 60 --warmup 10 --blocks 2 --min-block-s 0.1 $KSTAT_ARGS > gpurun_out/ks_$TAG.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/ks_$TAG/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows[:int("${KSTAT_ROWS:-16}")]:
    n = r['Name'].replace('void dlwpcs::','')[:95]
    print('%8.1f us x %6s  %s' % (float(r['AverageNs'])/1e3, r['Calls'], n))
PY
