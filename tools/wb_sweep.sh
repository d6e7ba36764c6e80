#!/bin/bash
# cost-model sweep of the batched weight gradient's plan (side build: DLWPCS_LIB_TAG=tune, -DDLWPCS_WB_TUNE_ENV=1); settings are
# pre-filtered on the CPU: `DLWPCS_LIB_TAG=tune DLWPCS_WB_COST=... pytest tests/test_wgrad_batch_plan.py` (every worker has work)
cd /root/repo
export DLWPCS_LIB_TAG=tune
run() { for r in 1 2 3; do env "$@" python tools/wb_bench.py --reps 40 2>/dev/null | grep wgrad_batch_kernel | awk '{printf "%s ", $2}'; done; echo; }
# fix,bpc,slab3,slab1,ld4,cfix   shipped: 3300,35,570,260,45,2000
for rep in 1 2; do
for c in ${COSTS:-"3300,35,570,260,45,2000"}; do echo "$c: $(run DLWPCS_WB_COST=$c)"; done
for seg in ${SEGS:-}; do echo "seg=$seg: $(run DLWPCS_WB_SEG=$seg)"; done
done
