#!/usr/bin/env python3
"""Overlap analysis of a rocprofv3 (rocpd / SQLite) kernel trace of bench.py: how much of the wall time of the steady-state
steps has an MFMA kernel (igemm / wgrad) in flight, how much only HBM-bound kernels, how much nothing.
Usage: python tools/timeline.py <results.db> [skip_fraction]"""
import sqlite3
import sys

db = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.45
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute('pragma table_info(kernels)').fetchall()]
print('columns:', cols)
scol = 'stream_id' if 'stream_id' in cols else ('queue_id' if 'queue_id' in cols else None)
q = 'select name, start, end%s from kernels order by start' % (', ' + scol if scol else '')
rows = c.execute(q).fetchall()
t0, t1 = rows[0][1], max(r[2] for r in rows)
lo = t0 + skip * (t1 - t0)  # skip warm-up / set-up
rows = [r for r in rows if r[1] >= lo]
is_mfma = lambda n: ('igemm_f32_kernel' in n) or ('winograd_f32_kernel' in n) or ('convt2x2_f32_kernel' in n) or ('wgrad_f32_kernel' in n) or ('wgrad_tr_kernel' in n) or ('wgrad_patch_kernel' in n)
ev = []
for r in rows:
    k = 1 if is_mfma(r[0]) else 0
    ev.append((r[1], 1, k))
    ev.append((r[2], -1, k))
ev.sort()
act = [0, 0]
last = ev[0][0]
tot = {'mfma': 0, 'mfma2+': 0, 'other_only': 0, 'idle': 0}
for t, d, k in ev:
    dt = t - last
    if act[1] > 0:
        tot['mfma'] += dt
        if act[1] > 1:
            tot['mfma2+'] += dt
    elif act[0] > 0:
        tot['other_only'] += dt
    else:
        tot['idle'] += dt
    act[k] += d
    last = t
span = ev[-1][0] - ev[0][0]
print('window %.3f ms, %d kernels' % (span / 1e6, len(rows)))
for k, v in tot.items():
    print('  %-11s %8.3f ms  %5.1f %%' % (k, v / 1e6, 100.0 * v / span))
if scol:
    streams = {}
    for r in rows:
        streams.setdefault(r[3], [0, 0])
        streams[r[3]][0] += 1
        streams[r[3]][1] += r[2] - r[1]
    for s, (n, d) in sorted(streams.items(), key=lambda kv: -kv[1][1]):
        print('  stream %s: %d kernels, busy %.3f ms (%.1f %% of window)' % (s, n, d / 1e6, 100.0 * d / span))
# per-kernel-name time when it was the ONLY thing running vs total
