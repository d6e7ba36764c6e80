#!/usr/bin/env python3
"""Per-position launch durations of one hipGraph-replayed bf16 training step from a rocprofv3 --kernel-trace directory.
usage: tools/step_positions.py <dir> [<dir2> ...]   (several directories: side-by-side columns)"""
import collections, csv, glob, re, sys


def load(d):
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'adam_fused' in r['Kernel_Name']]
    steps = [(a, b) for a, b in zip(idx[:-1], idx[1:]) if any('wgrad_bf16' in r['Kernel_Name'] for r in rows[a + 1:b + 1])]
    n = collections.Counter(b - a for a, b in steps).most_common(1)[0][0]
    steps = [s for s in steps if s[1] - s[0] == n][-20:]
    pos, names = collections.defaultdict(list), {}
    for a, b in steps:
        for k, r in enumerate(rows[a + 1:b + 1]):
            pos[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
            names[k] = re.sub(r'\(.*\)$', '', r['Kernel_Name']).replace('void ', '').replace('dlwpcs::', '')
    return [(sum(pos[k]) / len(pos[k]), names[k]) for k in sorted(pos)]


cols = [load(d) for d in sys.argv[1:]]
for k in range(max(len(c) for c in cols)):
    cells = ['%6.1f' % c[k][0] if k < len(c) else '      ' for c in cols]
    print('%2d %s  %s' % (k, ' '.join(cells), ' | '.join(c[k][1][:70] for c in cols if k < len(c))))
print('sum ' + ' '.join('%6.1f' % sum(x[0] for x in c) for c in cols))
