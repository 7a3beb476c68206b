"""Kernel sequence of the LAST inference forward of a rocprofv3 (rocpd) kernel trace of `tools/cfg_timing.py ... eval` on 16-bit storage: start
(us from the input conversion), duration, stream, kernel — which launches sit on the main stream, which beside it.
usage: eval_sequence.py results.db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,stream_id from kernels order by start").fetchall()
# last forward: from the last nchw3_to_padded4 to the end
idx = [i for i, r in enumerate(rows) if 'nchw3_to_padded4' in r[0]]
lo = idx[-1]
t0 = rows[lo][1]
for n, s, e, st in rows[lo:]:
    nm = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print('%9.1f %8.1f  s%d %s' % ((s - t0) / 1e3, (e - s) / 1e3, st % 10, nm[:80]))
