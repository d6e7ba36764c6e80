#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (CSV kernel trace) per kernel: launches, total, average, share."""
import collections
import csv
import sys


def main(path, out=None, skip_first_ms=0.0):
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict()
    for r in rows:
        k = r['Kernel_Name']
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = ['%6s %8s %12s %10s %10s %10s  %s' % ('share', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'kernel')]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append('%5.1f%% %8d %12.1f %10.2f %10.2f %10.2f  %s' % (100 * a[1] / tot, a[0], a[1], a[1] / a[0], a[2], a[3], k[:150]))
    lines.append('total kernel time: %.3f ms over %d dispatches' % (tot / 1e3, len(rows)))
    text = '\n'.join(lines)
    if out:
        open(out, 'w').write(text + '\n')
    print(text)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
