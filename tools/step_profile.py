#!/usr/bin/env python3
"""Per-kernel summary of one training step out of a rocprofv3 --kernel-trace CSV (a step ends at its optimizer launch).
usage: tools/step_profile.py <dir with *kernel_trace.csv> [--order]"""
import collections
import csv
import glob
import re
import sys


def clean(x):
    return re.sub(r'\(.*\)$', '', x).replace('void ', '').replace('dlwpcs::', '')


def main():
    trace = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # a training step ends with its optimizer launch; a rollout (none) starts with the weight packing
    ends = [i + 1 for i, r in enumerate(rows) if 'wb_reduce_kernel' in r['Kernel_Name'] or 'adam_fused_kernel' in r['Kernel_Name']]
    idx = ends if len(ends) > 2 else [i for i, r in enumerate(rows) if 'pack_batch_kernel' in r['Kernel_Name']]
    steps = [rows[a:b] for a, b in zip(idx[:-1], idx[1:])]
    n = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if len(s) == n][-30:]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for s in steps:
        for r in s:
            a = agg[clean(r['Kernel_Name'])]
            a[0] += 1
            a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot = sum(v[1] for v in agg.values())
    ns = len(steps)
    span = sum((int(s[-1]['End_Timestamp']) - int(s[0]['Start_Timestamp'])) / 1e3 for s in steps) / ns
    print('launches/step %d, kernel-busy %.1f us, span %.1f us (%d steps)' % (n, tot / ns, span, ns))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%8.1f %5.1f%% %5.1f %8.1f  %s' % (v[1] / ns, 100 * v[1] / tot, v[0] / ns, v[1] / v[0], k[:120]))
    if '--order' in sys.argv:
        print('--- order (last step)')
        for r in steps[-1]:
            print('%7.1f %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, clean(r['Kernel_Name'])[:110]))


if __name__ == '__main__':
    main()
