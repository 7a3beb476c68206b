#!/usr/bin/env python3
"""Print per-kernel PMC counter averages from a rocprofv3 rocpd database."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name order by kernel_name").fetchall() if 'kernel_name' in cols else []
if not rows:
    print(cols)
for k, n, v, cnt in rows:
    k = k.replace('(anonymous namespace)::', '')[:70]
    print('%-72s %-28s %16.1f (n=%d)' % (k, n, v, cnt))
