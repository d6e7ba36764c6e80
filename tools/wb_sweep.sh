cd /root/repo
run() { env "$@" python tools/wb_bench.py --reps 30 2>/dev/null | tail -3 | tr '\n' ' '; echo; }
echo base; run X=1
for seg in 10000 30000 40000; do echo seg=$seg; run DLWPCS_WB_SEG=$seg; done
# fix,bpc,slab3,slab1,ld4,cfix   defaults 3300,23,530,260,45,1200
for c in "3300,20,530,260,45,1200" "3300,26,530,260,45,1200" "2500,23,530,260,45,1200" "4200,23,530,260,45,1200" "3300,23,480,260,45,1200" "3300,23,580,260,45,1200" "3300,23,530,260,45,600" "3300,23,530,260,45,2000" "3300,23,530,200,45,1200" "3300,23,530,330,45,1200" "3300,23,530,260,30,1200" "3300,23,530,260,60,1200"; do echo cost=$c; run DLWPCS_WB_COST=$c; done
for w in 248 240; do echo workers=$w; run DLWPCS_WB_WORKERS=$w; done
