#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results.db: per-kernel stats CSV (profiles/) and, optionally, the kernel
sequence of one nav-step (between two gaa_pool launches).
usage: python tools/prof_summary.py results.db out.csv [--seq]"""
import csv, re, sqlite3, sys


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('at::native::', '').replace('void ', '')
    m = re.match(r'([A-Za-z_0-9:]+)(<[^(]{0,60})?', n)
    return (m.group(0) if m else n)[:80]


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    rows = c.execute('select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels '
                     'group by name order by 3 desc').fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for n, k, t, a, mn, mx in rows:
            w.writerow([n, k, t, '%.1f' % a, '%.2f' % (100 * t / tot), mn, mx])
    for n, k, t, a, mn, mx in rows[:14]:
        print('%-70s %6d calls %9.1f us  avg %7.1f us %5.1f%%' % (short(n)[:70], k, t / 1e3, a / 1e3, 100 * t / tot))
    if '--seq' in sys.argv:
        ks = c.execute('select name,start,end,grid_x,grid_y,grid_z from kernels order by start').fetchall()
        idx = [i for i, r in enumerate(ks) if 'gaa_pool' in r[0]]
        a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
        agg = {}
        for r in ks[a:b]:
            d = agg.setdefault(short(r[0]), [0, 0.0]); d[0] += 1; d[1] += (r[2] - r[1]) / 1e3
            print('%7.1f  %-80s %s' % ((r[2] - r[1]) / 1e3, short(r[0]), r[3:6]))
        print('--- one step: %d kernels, %.1f us busy, %.1f us span' % (b - a, sum(v[1] for v in agg.values()), (ks[b][1] - ks[a][1]) / 1e3))
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print('%7.1f us %3d x  %s' % (v[1], v[0], k))


if __name__ == '__main__':
    main()
