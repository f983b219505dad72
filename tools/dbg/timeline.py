#!/usr/bin/env python3
"""Timeline of one replayed nav-step from a rocprofv3 --kernel-trace results.db: start offset, duration, queue, name.
usage: python tools/dbg/timeline.py results.db > timeline.txt"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print('# columns:', cols)
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
ks = c.execute('select name,start,end,grid_x,%s from kernels order by start' % (qcol or '0')).fetchall()
idx = [i for i, r in enumerate(ks) if 'gaa_pool' in r[0]]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
# a step starts at the first kernel after the previous step's last one: rotate so that the print starts at the panorama's first kernel
t0 = ks[a][1]
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('at::native::', '').replace('void ', '')
    return re.sub(r'\(.*', '', n)[:86]
qs = sorted(set(r[4] for r in ks[a:b]))
busy = dict((q, 0.0) for q in qs)
for r in ks[a:b]:
    q = qs.index(r[4])
    busy[r[4]] += (r[2] - r[1]) / 1e3
    print('%8.1f %7.1f  %s%-3d %7d  %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, '    ' * q, q, r[3], short(r[0])))
print('# span %.1f us; busy per queue:' % ((ks[b][1] - t0) / 1e3), ' '.join('%d:%.1f' % (qs.index(q), v) for q, v in busy.items()))
