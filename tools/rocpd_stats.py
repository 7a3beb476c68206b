#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd / SQLite) kernel trace: per-kernel calls, total, average, share.
Usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
                 "order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
lines = ['| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
for name, n, s, a, mn, mx in rows:
    name = name.replace('(anonymous namespace)::', '')
    if len(name) > 90:
        name = name[:87] + '...'
    lines.append('| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.2f |' % (name, n, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
lines.append('| **total** | %d | %.3f | | | | 100 |' % (sum(r[1] for r in rows), tot / 1e6))
out = '\n'.join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(out + '\n')
