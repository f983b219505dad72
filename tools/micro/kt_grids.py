"""Per (kernel, grid) launch statistics from a rocprofv3 kernel trace: python3 kt_grids.py <results.db> <substring>"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute('select s.kernel_name, d.end - d.start, d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id = s.id' % (kd, ks)).fetchall()
agg = {}
for n, t, g, w in rows:
    if pat in n:
        key = (n[n.find(pat):n.find('(', n.find(pat))][:60], g // max(w, 1))
        agg.setdefault(key, []).append(t)
for (n, wg), ts in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    ts.sort()
    print('%-62s wgs %5d  calls %4d  median %8.1f us  total %8.2f ms' % (n, wg, len(ts), ts[len(ts) // 2] / 1e3, sum(ts) / 1e6))
