#!/bin/bash
# cost-model sweep of the batched weight gradient's plan (side build: DLWPCS_LIB_TAG=tune, -DDLWPCS_WB_TUNE_ENV=1)
cd /root/repo
export DLWPCS_LIB_TAG=tune
run() { env "$@" python tools/wb_bench.py --reps 30 2>/dev/null | grep wgrad_batch_kernel | awk '{print $2}'; }
echo "base $(run X=1) $(run X=1)"
# fix,bpc,slab3,slab1,ld4,cfix   defaults 3300,23,530,260,45,1200
for fix in 500 1200 2000 3300; do for bpc in 23 35 60; do echo "fix=$fix bpc=$bpc: $(run DLWPCS_WB_COST=$fix,$bpc,530,260,45,1200)"; done; done
for slab in 450 490 570 620; do echo "slab3=$slab: $(run DLWPCS_WB_COST=3300,23,$slab,260,45,1200)"; done
for cf in 600 2000 3000; do echo "cfix=$cf: $(run DLWPCS_WB_COST=3300,23,530,260,45,$cf)"; done
for s1 in 200 330 420; do echo "slab1=$s1: $(run DLWPCS_WB_COST=3300,23,530,$s1,45,1200)"; done
for seg in 5000 10000 30000 40000; do echo "seg=$seg: $(run DLWPCS_WB_SEG=$seg)"; done
