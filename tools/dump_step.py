#!/usr/bin/env python3
"""Chronological dump of ONE steady-state step of a rocprofv3 (rocpd) kernel trace of bench.py: start / end (us from the step's
start), stream, kernel — to see which stream the step's end waits for and where a stream sits idle.
usage: dump_step.py results.db [step_index]"""
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,stream_id from kernels order by start").fetchall()
adam = [r for r in rows if 'adam_kernel' in r[0]]
step = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lo, hi = adam[step - 1][2], adam[step][2]
ks = [r for r in rows if r[1] >= lo and r[2] <= hi]
streams = sorted({r[3] for r in ks}, key=lambda s: -sum(1 for r in ks if r[3] == s))
col = {s: i for i, s in enumerate(streams)}
print('step %d: %.1f us, streams %s' % (step, (hi - lo) / 1e3, streams))
last_end = {}
for n, s, e, st in ks:
    nm = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    gap = (s - last_end[st]) / 1e3 if st in last_end else 0.0
    last_end[st] = e
    print('%9.1f %9.1f %7.1f  gap %7.1f  s%d %s%s' % ((s - lo) / 1e3, (e - lo) / 1e3, (e - s) / 1e3, gap, col[st], '    ' * col[st], nm[:70]))
