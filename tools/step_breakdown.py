#!/usr/bin/env python3
"""Per-step breakdown of a rocprofv3 (rocpd) kernel trace of bench.py: wall time with an MFMA kernel in flight, with only
HBM-bound kernels in flight, idle; per kernel family its total and its EXCLUSIVE time (no MFMA kernel beside it).
usage: step_breakdown.py results.db [step_index]"""
import collections
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,stream_id from kernels order by start").fetchall()
adam = [r for r in rows if 'adam_kernel' in r[0]]
step = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lo, hi = adam[step - 1][2], adam[step][2]
ks = [r for r in rows if r[1] >= lo and r[2] <= hi]
ism = lambda n: ('igemm_f32_kernel' in n) or ('winograd_f32_kernel' in n) or ('convt2x2_f32_kernel' in n) or ('wgrad_f32_kernel' in n) or ('wgrad_tr_kernel' in n) or ('wgrad_patch_kernel' in n) or ('conv3x3_wres16_kernel' in n) or ('convt2x2_b16_kernel' in n) or ('stem7x7_' in n) or ('head16_tail_eval_kernel' in n)
mf = sorted([(r[1], r[2]) for r in ks if ism(r[0])])
merged = []
for s, e in mf:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
def excl(s, e):
    cov = 0
    for a, b in merged:
        if b <= s:
            continue
        if a >= e:
            break
        cov += min(e, b) - max(s, a)
    return (e - s) - cov
mfma_t = sum(b - a for a, b in merged)
oth = sorted([(r[1], r[2]) for r in ks if not ism(r[0])])
m2 = []
for s, e in oth:
    if m2 and s <= m2[-1][1]:
        m2[-1][1] = max(m2[-1][1], e)
    else:
        m2.append([s, e])
other_only = sum(excl(a, b) for a, b in m2)
span = hi - lo
print('step %d: %.3f ms, %d kernels; MFMA kernel in flight %.3f ms (%.1f %%), only other kernels %.3f ms, idle %.3f ms' %
      (step, span / 1e6, len(ks), mfma_t / 1e6, 100.0 * mfma_t / span, other_only / 1e6, (span - mfma_t - other_only) / 1e6))
tot = collections.defaultdict(lambda: [0, 0, 0])
for r in ks:
    n = r[0].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    t = tot[n]
    t[0] += 1
    t[1] += r[2] - r[1]
    if not ism(r[0]):
        t[2] += excl(r[1], r[2])
print('%-44s %5s %9s %9s' % ('kernel', 'calls', 'total us', 'excl us'))
for n, (k, d, x) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:28]:
    print('%-44s %5d %9.1f %9.1f' % (n[:44], k, d / 1e3, x / 1e3))
