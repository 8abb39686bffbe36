#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV into per-kernel averages (small enough to commit).

usage: pmc_summary.py <dir-with-*counter_collection.csv> <out.csv>
Applies no correction itself; DESIGN.md states the gfx950 rule (FETCH_SIZE is in KiB-units of 64-B requests and
under-reports wide coalesced streaming reads by 2x, MI355X_MICROARCH.md section HBM)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(src, out):
    files = glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True)
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get('Kernel_Name', '?')
                k = k.replace('(anonymous namespace)::', '').replace('void ', '')
                k = k.split('(')[0][:120]
                c = row.get('Counter_Name')
                v = float(row.get('Counter_Value', 0) or 0)
                a = agg[k][c]
                a[0] += v
                a[1] += 1
    with open(out, 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['kernel', 'counter', 'dispatches', 'mean_per_dispatch', 'total'])
        for k in sorted(agg, key=lambda k: -sum(v[0] for v in agg[k].values())):
            for c, (tot, n) in sorted(agg[k].items()):
                w.writerow([k, c, n, tot / max(n, 1), tot])
    print('wrote', out, 'from', len(files), 'file(s)')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
