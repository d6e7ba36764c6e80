#!/bin/bash
# cost-model sweep of the batched weight gradient's plan (side build: DLWPCS_LIB_TAG=tune, -DDLWPCS_WB_TUNE_ENV=1)
cd /root/repo
export DLWPCS_LIB_TAG=tune
run() { a=$(env "$@" python tools/wb_bench.py --reps 40 2>/dev/null | grep wgrad_batch_kernel | awk '{print $2}'); b=$(env "$@" python tools/wb_bench.py --reps 40 2>/dev/null | grep wgrad_batch_kernel | awk '{print $2}'); echo "$a $b"; }
echo "base $(run X=1)"
# fix,bpc,slab3,slab1,ld4,cfix   defaults 3300,23,530,260,45,1200
for c in "3300,35,530,260,45,1200" "2000,35,530,260,45,1200" "2000,23,530,260,45,1200" "3300,23,530,260,45,2000" "3300,35,530,260,45,2000" "2000,35,530,260,45,2000" "2600,30,530,260,45,1600" "2600,30,570,260,45,1600" "3300,35,570,260,45,2000" "2000,35,570,260,45,2000" "1600,30,530,260,45,2400" "2600,40,530,260,45,2000" "3300,30,550,260,45,2400"; do echo "$c: $(run DLWPCS_WB_COST=$c)"; done
echo "base $(run X=1)"
