#!/bin/bash
# SQ / LDS counters of the batched weight gradient alone (tools/wb_bench.py): where the waves wait
# usage: tools/wb_sq.sh <label> [ENV=VAL ...]
LABEL=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/wbsq_$LABEL; rm -rf $OUT; mkdir -p $OUT
G1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS"
G2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"
i=0
for g in "$G1" "$G2"; do
  env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $g -d $OUT/p$i -o p --output-format csv -- python tools/wb_bench.py --reps 3 ${WB_ARGS:-} > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python - "$LABEL" "$OUT" <<'PY'
import csv, glob, sys, collections
label, out = sys.argv[1:3]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('%s/**/*counter_collection.csv' % out, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'wgrad_batch_kernel' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print('== %s (per launch)' % label)
for k in sorted(acc):
    print('  %-28s %14.0f' % (k, acc[k][0] / max(acc[k][1], 1)))
PY
