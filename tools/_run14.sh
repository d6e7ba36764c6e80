mkdir -p gpurun_out
python bench.py > gpurun_out/r5_bench14.json 2> gpurun_out/r5_bench14.err
tail -c 3000 gpurun_out/r5_bench14.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench14.json').read().strip().split('\n')[-1])
print('value',d['value'],'ms',d['ms_per_step'])
rf=d['roofline']; print({k:rf[k] for k in ('kernel','bound','frac','avg_launch_us','traffic','traffic_vs_algorithmic','mfma_busy','pmc_missing') if k in rf})
for k,v in rf['per_kernel'].items(): print('  ',k[:100],v)
print('dp_form',d.get('dp_form'))
print('f32',{k:d['f32'][k] for k in ('value','ms_per_step')}, {k:d['f32']['roofline'][k] for k in ('kernel','bound','frac','avg_launch_us','traffic_vs_algorithmic','mfma_busy','pmc_missing')})
for k,v in d['configs'].items(): print(k, v['value'], v['ms_per_step'], v.get('roofline'))
print('cpu',d['cpu_baseline']['value'],d['cpu_baseline']['cores'])
PY
