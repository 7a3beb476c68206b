#!/usr/bin/env python3
"""Per kernel: SALU / VALU / LDS instructions per MFMA, parked and issue-stalled share of the wave cycles — from a rocprofv3 pass with
--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU.
usage: insn_mix.py results.db [cycles per MFMA: 32 for 32x32x16 bf16, 64 for 32x32x2 f32]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cyc = float(sys.argv[2]) if len(sys.argv) > 2 else 32.0
rows = {}
for k, name, s, n in c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace(", ", ",")
    rows.setdefault(k, {})[name] = s / n
out = []
for k, v in rows.items():
    m = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / cyc
    if m > 0:
        out.append((v["SQ_WAVE_CYCLES"], k, v["SQ_INSTS_SALU"] / m, v["SQ_INSTS_VALU"] / m, v["SQ_INSTS_LDS"] / m,
                    100 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
print("%-48s %8s %8s %8s %7s %7s" % ("kernel", "SALU/MF", "VALU/MF", "LDS/MF", "wait%", "stall%"))
for o in sorted(out, reverse=True):
    print("%-48s %8.1f %8.1f %8.1f %7.0f %7.0f" % o[1:])
