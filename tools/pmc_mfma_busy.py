#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from ONE rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE (and, when present,
SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY).  usage: pmc_mfma_busy.py results.db [min share of the MFMA cycles]
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   — cycles the matrix pipes were busy / cycles they existed
    (GRBM_GUI_ACTIVE comes back summed over the 8 XCDs: calibrated on csrc/mfma_probe.hip's pure-MFMA kernel, which reads 0.99 this way)
    (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per v_mfma_f32_32x32x16, 64 per v_mfma_f32_32x32x2_f32; GRBM_GUI_ACTIVE
    = the kernel's cycles at the clock it actually ran at — so this is utilisation at the ACTUAL clock, not against the 2.4 GHz of the
    nominal peak)
  parked / issue-stalled / issuing = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (quad-cycles each)"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.005
rows = c.execute("select kernel_name, counter_name, avg(value), sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
by = {}
for k, cn, avg, tot, cnt in rows:
    name = k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].replace(', ', ',')
    by.setdefault(name, {})[cn] = (avg, tot, cnt)
tot_mfma = sum(d.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0, 0))[1] for d in by.values()) or 1.0
print('| kernel | launches | avg kernel cycles (GRBM_GUI_ACTIVE / 8) | matrix pipe busy | waves parked | issue-stalled | issuing | share of all MFMA cycles |')
print('|---|---|---|---|---|---|---|---|')
for name, d in sorted(by.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0, 0))[1]):
    m = d.get('SQ_VALU_MFMA_BUSY_CYCLES')
    g = d.get('GRBM_GUI_ACTIVE')
    if not m or not g or m[1] / tot_mfma < floor:
        continue
    wc = d.get('SQ_WAVE_CYCLES', (0, 0, 0))[0] or float('nan')
    f = lambda key: d.get(key, (float('nan'), 0, 0))[0] / wc
    print('| `%s` | %d | %.0f | %.3f | %.2f | %.2f | %.2f | %.3f |' % (name, m[2], g[0] / 8.0, m[0] / (1024.0 * g[0] / 8.0), f('SQ_WAIT_ANY'), f('SQ_WAIT_INST_ANY'),
                                                                 f('SQ_ACTIVE_INST_ANY'), m[1] / tot_mfma))
