#!/usr/bin/env python3
"""Tabulate gpurun_out/g_<tile>.log files written by tools/bench_gemm.py under DSRG_GEMM_TILE=<tile>."""
import glob, os, re, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
names = sys.argv[1:] or ['auto', '64x64x2', '64x64x3', '64x128x2', '64x128x3', '128x128x2', '128x128x3', '256x128x2', '256x128x3', '256x192x2']
rows = {}
for n in names:
    for l in open(os.path.join(root, 'g_%s.log' % n)):
        m = re.match(r'(.{44})\s+([\d.]+) us', l)
        if m:
            rows.setdefault(m.group(1).strip(), {})[n] = float(m.group(2))
print('%-44s' % 'shape' + ''.join('%10s' % n for n in names))
for k, v in rows.items():
    print('%-44s' % k + ''.join('%10.1f' % v.get(n, 0) for n in names))
