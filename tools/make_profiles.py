#!/usr/bin/env python3
"""
Condense the rocprofv3 outputs of `python bench.py` into the committed summaries under profiles/.

Inputs (written on the GPU box by tools/gpu_profile.sh, merged back under gpurun_out/):
  gpurun_out/prof_<tag>/r_kernel_trace.csv, r_kernel_stats.csv   rocprofv3 --kernel-trace --stats -- python bench.py ...
  gpurun_out/bench_<tag>.json                                    the bench line of that same command (un-profiled run)
  gpurun_out/pmc_<tag>.json                                      bench.py --pmc-out: per-kernel FETCH_SIZE / WRITE_SIZE /
                                                                 SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE records (live passes)
Outputs:
  profiles/<tag>_bench_unet2_b32_kernel_summary.txt   per-kernel time per hipGraph-replayed step, both dtypes
  profiles/<tag>_bench_unet2_b32_kernel_stats.csv     rocprofv3's own --stats table
  profiles/<tag>_bench_unet2_b32_benchline.json       the JSON line
  profiles/<tag>_bench_unet2_b32_pmc.json             PMC records (bench.py falls back to the newest of these)
  profiles/<tag>_bench_unet2_b32_pmc.txt              the same as a table (traffic vs algorithmic bytes, MFMA busy)
usage: tools/make_profiles.py [tag]      (default r02)
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r03'
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
    out = ['%s: %d steps, %d launches/step, kernel-busy %.1f us/step, wall span %.1f us/step'
           % (title, nsteps, round(len(sel) / nsteps), tot / nsteps, span / nsteps),
           '%10s %6s %7s %9s  %s' % ('us/step', '%', 'n/step', 'avg_us', 'kernel')]
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append('%10.1f %6.1f %7.1f %9.1f  %s' % (v[1] / nsteps, 100 * v[1] / tot, v[0] / nsteps, v[1] / v[0], n[:120]))
    return '\n'.join(out)


def step_summaries(pdir, modes):
    """modes: [(is_bf16 | None, title)]; see the step delimiters below"""
    trace = glob.glob(os.path.join(pdir, '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # a training step ENDS with its optimizer launch (the weight-gradient reduction + Adam, or the Adam kernel); a rollout (no
    # optimizer) starts at its weight-packing launch
    ends = [i + 1 for i, r in enumerate(rows) if 'wb_reduce_kernel' in r['Kernel_Name'] or 'adam_fused_kernel' in r['Kernel_Name']]
    # (a step with ONE batched weight-gradient launch is cut there instead: a model whose layers are applied twice -- unet2x2 -- runs
    # two reduction launches and an optimizer launch behind it; the cut is rotated against the step, the per-step sums are not)
    wb = [i + 1 for i, r in enumerate(rows) if 'wgrad_batch_kernel' in r['Kernel_Name']]
    if len(wb) > 2:
        ends = wb
    idx = ends if len(ends) > 2 else [i for i, r in enumerate(rows) if 'pack_batch_kernel' in r['Kernel_Name']]
    steps = []
    for a, b in zip(idx[:-1], idx[1:]):
        seg = rows[a:b]
        bf = any('unsigned short' in r['Kernel_Name'] or 'pw_' in r['Kernel_Name'] for r in seg)      # (both modes batch their weight gradients)
        steps.append((a, b, bf))
    txt = []
    for want, title in modes:
        st = [s for s in steps if want is None or s[2] == want]
        if not st:
            continue
        # hipGraph-replayed steps all have the same launch count: keep the most common length (drops warm-up / eager steps)
        lens = collections.Counter(s[1] - s[0] for s in st)
        common = lens.most_common(1)[0][0]
        st = [s for s in st if s[1] - s[0] == common][-40:]
        sel = [r for s in st for r in rows[s[0]:s[1]]]
        txt.append(summarize(sel, len(st), title))
    return txt


def main():
    pdir = os.path.join(R, 'gpurun_out', 'prof_%s' % TAG)
    head = 'rocprofv3 --kernel-trace --stats of `python bench.py` on 1 MI355X, '
    txt = step_summaries(pdir, ((True, head + 'bf16 mode (headline), hipGraph-replayed steps'),
                                (False, head + 'f32 mode (companion), hipGraph-replayed steps')))
    open(PRE + '_kernel_summary.txt', 'w').write('\n\n'.join(txt) + '\n')
    # BASELINE configs 2 and 5: their own traces (tools/gpu_profile.sh)
    for wl, name, title in (('encoder6', 'encoder6_b32', '`python bench.py --workload encoder6 --channels 7 --dtype f32` (BASELINE config 2), '
                                                          'hipGraph-replayed steps'),
                            ('rollout', 'rollout_b32', '`python bench.py --workload rollout` (BASELINE config 5: one step = one 40-step '
                                                        'rollout = 20 forward passes), eager launches'),
                            ('unet2x2', 'unet2x2_b32', '`python bench.py --workload unet2x2` (the reference scripts\' production model: '
                                                        'integration_steps 2, solar + constants inputs), hipGraph-replayed steps')):
        d2 = os.path.join(R, 'gpurun_out', 'prof_%s_%s' % (TAG, wl))
        if glob.glob(os.path.join(d2, '**', '*kernel_trace.csv'), recursive=True):
            t2 = step_summaries(d2, ((None, 'rocprofv3 --kernel-trace --stats of ' + title),))
            pre2 = os.path.join(R, 'profiles', '%s_bench_%s' % (TAG, name))
            open(pre2 + '_kernel_summary.txt', 'w').write('\n\n'.join(t2) + '\n')
            st2 = glob.glob(os.path.join(d2, '**', '*kernel_stats.csv'), recursive=True)
            if st2:
                shutil.copy(st2[0], pre2 + '_kernel_stats.csv')
    stats = glob.glob(os.path.join(pdir, '**', '*kernel_stats.csv'), recursive=True)
    if stats:
        shutil.copy(stats[0], PRE + '_kernel_stats.csv')
    shutil.copy(os.path.join(R, 'gpurun_out', 'bench_%s.json' % TAG), PRE + '_benchline.json')
    pmc = os.path.join(R, 'gpurun_out', 'pmc_%s.json' % TAG)
    if os.path.exists(pmc):
        doc = json.load(open(pmc))
        json.dump(doc, open(PRE + '_pmc.json', 'w'), indent=1, sort_keys=True)
        bench = json.load(open(os.path.join(R, 'gpurun_out', 'bench_%s.json' % TAG)))
        lines = ['Per-launch PMC records of `python bench.py` (live rocprofv3 passes, one counter group per pass: %s).' % doc.get('_source', ''),
                 'traffic = FETCH_SIZE x 2 (MI355X_MICROARCH.md: gfx950 rocprofv3 reports half of wide coalesced reads) + WRITE_SIZE;',
                 'mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCC x 1024 SIMDs)  (the profiler\'s MfmaUtil).', '']
        for key in sorted(k for k in doc if not k.startswith('_')):
            per = (bench if bench.get('dtype') == key.split('/')[1] else bench.get(key.split('/')[1], {})).get('roofline', {}).get('per_kernel', {})
            lines.append('== %s' % key)
            lines.append('%-84s %8s %10s %10s %10s %9s' % ('kernel', 'launches', 'fetchx2 KB', 'write KB', 'traffic MB', 'mfma_busy'))
            for k, v in sorted(doc[key].items(), key=lambda kv: -(kv[1].get('traffic') or 0) * (kv[1].get('launches') or 1)):
                lines.append('%-84s %8s %10s %10s %10s %9s' % (k[:84], v.get('launches', ''), v.get('fetch_kb_x2', ''),
                                                                 v.get('write_kb', ''),
                                                                 '%.1f' % (v['traffic'] / 1e6) if v.get('traffic') else '',
                                                                 v.get('mfma_busy', '')))
            lines.append('')
        open(PRE + '_pmc.txt', 'w').write('\n'.join(lines) + '\n')
    print(txt[0][:2400])


if __name__ == '__main__':
    main()
