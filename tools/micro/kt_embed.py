"""Summarise a rocprofv3 kernel trace of bench_embed_bwd.py: median kernel time per configuration (23 calls each)."""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute('select s.kernel_name, d.start, d.end, d.grid_size_x from %s d join %s s on d.kernel_id = s.id order by d.start' % (kd, ks)).fetchall()
emb = [(e - s, g) for n, s, e, g in rows if 'embed_ln_bwd' in n]
out = []
for i in range(0, len(emb), 23):
    d = sorted(x[0] for x in emb[i:i + 23])
    out.append('%d:%.0f' % (emb[i][1] // 256, d[len(d) // 2] / 1e3))
print(' '.join(out))
