#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 --pmc results.db -> CSV.
usage: python tools/pmc_summary.py results.db out.csv"""
import csv, sqlite3, sys


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = c.execute('select kernel_name, counter_name, grid_size, count(*), avg(value), avg(end-start) from counters_collection '
                     'group by kernel_name, counter_name, grid_size order by kernel_name, grid_size').fetchall()
    with open(sys.argv[2], 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Kernel', 'Counter', 'GridSize', 'Dispatches', 'AvgValue', 'AvgDurationNs'])
        for r in rows:
            w.writerow([r[0], r[1], r[2], r[3], '%.1f' % r[4], '%.1f' % r[5]])
    for r in rows:
        if 'gemm_nt' in r[0] or 'attention' in r[0]:
            print('%-74s %-14s grid %7d n=%3d avg=%14.1f  %7.1f us' % (r[0].replace('(anonymous namespace)::', '')[5:79], r[1], r[2], r[3], r[4], r[5] / 1e3))


if __name__ == '__main__':
    main()
