#!/usr/bin/env python3
"""Per-kernel SQ counter sums from one rocprofv3 --pmc pass (rocpd DB): where the waves' cycles go.
usage: pmc_sq.py results.db [name-filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
rows = {}
for k, name, s, n in c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
    k = k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].replace(', ', ',')
    rows.setdefault(k, {})[name] = (s, n)
names = sorted({n for v in rows.values() for n in v})
print('kernel | launches | ' + ' | '.join(names))
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', (0, 0))[0]):
    if flt not in k:
        continue
    n = max(x[1] for x in v.values())
    wc = v.get('SQ_WAVE_CYCLES', (0, 1))[0] or 1
    cells = []
    for nm in names:
        s = v.get(nm, (0, 0))[0]
        cells.append('%.3g (%.0f%%)' % (s / n, 100.0 * s / wc) if nm.startswith('SQ_WAIT') or nm.startswith('SQ_ACTIVE') or nm == 'SQ_VALU_MFMA_BUSY_CYCLES' else '%.3g' % (s / n))
    print('%s | %d | %s' % (k, n, ' | '.join(cells)))
