for m1 in 1.0 1.3 1.6 2.0; do for m3 in 1.0 1.1 1.2; do
echo -n "m3=$m3 m1=$m1: "; DLWPCS_WB_COST_F32=66,1200,$m3,$m1 timeout 120 python tools/probe_wb_f32.py --all --mask 2>&1 | grep " us"
done; done
