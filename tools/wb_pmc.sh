#!/bin/bash
# time + HBM-side counters of the batched weight gradient alone (tools/wb_bench.py), optionally of a side build:
# usage: tools/wb_pmc.sh <label> [ENV=VAL ...]      -> one line: us, FETCH_SIZE / WRITE_SIZE per launch (raw counter units: KiB?) 
LABEL=$1; shift
export TMPDIR=/tmp
T=$(env "$@" python tools/wb_bench.py --reps 30 2>/dev/null | grep wgrad_batch_kernel | awk '{print $2}')
OUT=gpurun_out/wbpmc_$LABEL
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o p --output-format csv -- python tools/wb_bench.py --reps 3 > $OUT/$c.log 2>&1
done
python - "$LABEL" "$T" "$OUT" <<'PY'
import csv, glob, sys
label, t, out = sys.argv[1:4]
vals = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    fs = glob.glob('%s/%s/**/*counter_collection.csv' % (out, c), recursive=True)
    tot, n = 0.0, 0
    for f in fs:
        for r in csv.DictReader(open(f)):
            if 'wgrad_batch_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c:
                tot += float(r['Counter_Value']); n += 1
    vals[c] = tot / n if n else float('nan')
# counter unit: KiB per the guide? print raw and the bench's convention (bytes = value * 1024? see bench.py) -- raw here
print('%-12s %8s us   FETCH_SIZE %12.0f   WRITE_SIZE %12.0f   (raw per launch)' % (label, t, vals['FETCH_SIZE'], vals['WRITE_SIZE']))
PY
