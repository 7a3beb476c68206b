#!/usr/bin/env python3
"""Per-kernel means of the counters of one or more rocprofv3 --pmc passes (results.db files): what a kernel's waves did with their
cycles.  usage: pmc_counters.py <substring of the kernel name> pass1.db [pass2.db ...]
Units per /opt/skills/guides/MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES in quad-cycles,
SQ_VALU_MFMA_BUSY_CYCLES in cycles; FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE x2 for wide coalesced reads on gfx950)."""
import sqlite3, sys
pat = sys.argv[1]
for db in sys.argv[2:]:
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    by = {}
    for k, cn, avg, cnt in rows:
        if pat in k:
            by.setdefault(k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0], {})[cn] = (avg, cnt)
    for k, d in sorted(by.items()):
        print(k)
        for cn, (avg, cnt) in sorted(d.items()):
            print('   %-28s %16.1f  (%d launches)' % (cn, avg, cnt))
