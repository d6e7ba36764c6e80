#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the batched weight gradient, one unet2 layer at a time (tools/wb_bench.py --layers i)
cd /root/repo
export TMPDIR=/tmp
for l in 0 1 2 3 4 5 6 7 8 9 10; do
  OUT=gpurun_out/wbpmc_layer$l; rm -rf $OUT; mkdir -p $OUT
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o p --output-format csv -- python tools/wb_bench.py --reps 3 --layers $l > $OUT/$c.log 2>&1
  done
  python - $l $OUT <<'PY'
import csv, glob, sys
l, out = sys.argv[1:3]
vals = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    tot, n = 0.0, 0
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (out, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'wgrad_batch_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c:
                tot += float(r['Counter_Value']); n += 1
    vals[c] = tot / n if n else float('nan')
print('layer %2s  FETCH x2 %7.1f MB   WRITE %6.1f MB' % (l, 2 * vals['FETCH_SIZE'] * 1024 / 1e6, vals['WRITE_SIZE'] * 1024 / 1e6))
PY
done
