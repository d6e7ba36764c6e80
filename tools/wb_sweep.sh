#!/bin/bash
# cost-model sweep of the batched weight gradient's plan (side build: DLWPCS_LIB_TAG=tune, -DDLWPCS_WB_TUNE_ENV=1)
cd /root/repo
export DLWPCS_LIB_TAG=tune
run() { for r in 1 2 3; do env "$@" python tools/wb_bench.py --reps 40 2>/dev/null | grep wgrad_batch_kernel | awk '{printf "%s ", $2}'; done; echo; }
# fix,bpc,slab3,slab1,ld4,cfix   defaults 3300,23,530,260,45,1200
for rep in 1 2; do
for c in "3300,23,530,260,45,1200" "3300,35,530,260,45,1200" "3300,35,530,260,45,2000" "3300,35,570,260,45,2000" "3000,30,530,260,45,1600"; do echo "$c: $(run DLWPCS_WB_COST=$c)"; done
done
echo "== step"; bash tools/ab.sh DLWPCS_WB_COST=3300,23,530,260,45,1200 -- DLWPCS_WB_COST=3300,35,530,260,45,1200 -- DLWPCS_WB_COST=3300,35,530,260,45,2000 -- DLWPCS_WB_COST=3300,23,530,260,45,1200 -- DLWPCS_WB_COST=3300,35,530,260,45,1200 --
