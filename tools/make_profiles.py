#!/usr/bin/env python3
"""
Condense the rocprofv3 outputs of `python bench.py` into the committed summaries under profiles/.

  gpurun_out/prof_r1/r1_kernel_trace.csv   <- rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py
  gpurun_out/pmc_FETCH_SIZE|WRITE_SIZE/    <- rocprofv3 --kernel-trace --pmc <counter> ... -- python bench.py --steps 3
                                              --warmup 3 --no-cpu-baseline --no-roofline --no-graphs   (one counter per pass)
usage: tools/make_profiles.py [round_tag]      (default r01)
"""
import collections
import csv
import os
import re
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r01'
PRE = os.path.join(R, 'profiles', '%s_bench_unet2_b32' % TAG)


def clean(n):
    return re.sub(r'\(.*\)$', '', n).replace('void ', '').replace('dlwpcs::', '')


def summarize(sel, nsteps, title):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in sel:
        a = agg[clean(r['Kernel_Name'])]
        a[0] += 1
        a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot = sum(v[1] for v in agg.values())
    span = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e3
    out = ['%s: %d steps, kernel-busy %.1f us/step, wall span %.1f us/step' % (title, nsteps, tot / nsteps, span / nsteps),
           '%10s %6s %7s %9s  %s' % ('us/step', '%', 'n/step', 'avg_us', 'kernel')]
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append('%10.1f %6.1f %7.1f %9.1f  %s' % (v[1] / nsteps, 100 * v[1] / tot, v[0] / nsteps, v[1] / v[0], n[:120]))
    return '\n'.join(out)


def main():
    rows = list(csv.DictReader(open(os.path.join(R, 'gpurun_out/prof_r1/r1_kernel_trace.csv'))))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'adam_fused_kernel' in r['Kernel_Name']]
    steps = []
    for a, b in zip(idx[:-1], idx[1:]):           # a step = the kernels after one optimizer launch up to the next one
        seg = rows[a + 1:b + 1]
        bf = any('unsigned short' in r['Kernel_Name'] or 'wgrad_bf16' in r['Kernel_Name'] for r in seg)
        steps.append((a + 1, b + 1, bf))
    txt = []
    for want, title in ((True, 'bf16 mode (headline)'), (False, 'f32 mode (companion)')):
        st = [s for s in steps if s[2] == want][:-4][-40:]      # drop the eager roofline-pass steps at the end of each mode
        sel = rows[st[0][0]:st[-1][1]]
        txt.append(summarize(sel, len(st), 'rocprofv3 --kernel-trace --stats of `python bench.py` on 1 MI355X, ' + title +
                             ', hipGraph-replayed steps'))
    open(PRE + '_kernel_summary.txt', 'w').write('\n\n'.join(txt) + '\n')
    shutil.copy(os.path.join(R, 'gpurun_out/prof_r1/r1_kernel_stats.csv'), PRE + '_kernel_stats.csv')
    shutil.copy(os.path.join(R, 'gpurun_out/bench_default.json'), PRE + '_benchline.json')

    pm = {}
    keep = ('conv_mfma', 'wgrad', 'pw_', 'pad_bwd_src', 'pad_ring_fix', 'pack_batch', 'avgpool', 'mse_stage1', 'adam')
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        for r in csv.DictReader(open(os.path.join(R, 'gpurun_out/pmc_%s/p_counter_collection.csv' % c))):
            k = clean(r['Kernel_Name'])
            if any(x in k for x in keep):
                a = pm.setdefault(k, {}).setdefault(c, [0, 0.0])
                a[0] += 1
                a[1] += float(r['Counter_Value'])
    lines = ['HBM-side traffic per launch from rocprofv3 PMC passes (one counter per pass: `rocprofv3 --kernel-trace --pmc',
             'FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-graphs`, both dtypes',
             'in one run, 1 MI355X).  Units: KB as reported.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly',
             '1/2 of the bytes of wide (16 B/lane) coalesced streaming reads -> "x2" column; WRITE_SIZE is uncalibrated (as is).  The',
             '256 MiB Infinity Cache absorbs re-reads of tensors < ~100 MB: upper bounds on DRAM traffic, not over-fetch evidence.', '',
             '%-84s %8s %12s %12s %12s' % ('kernel', 'launches', 'FETCH KB', 'FETCH x2 KB', 'WRITE KB')]
    for k, v in sorted(pm.items(), key=lambda kv: -kv[1].get('FETCH_SIZE', [0, 0])[1]):
        f, w = v.get('FETCH_SIZE', [1, 0]), v.get('WRITE_SIZE', [1, 0])
        lines.append('%-84s %8d %12.0f %12.0f %12.0f' % (k[:84], f[0], f[1] / max(f[0], 1), 2 * f[1] / max(f[0], 1),
                                                          w[1] / max(w[0], 1)))
    open(PRE + '_hbm_pmc.txt', 'w').write('\n'.join(lines) + '\n')
    print(txt[0][:1800])
    print('\n'.join(lines[6:12]))


if __name__ == '__main__':
    main()
