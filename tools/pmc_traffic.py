#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on
gfx950).  Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: the counters are in KiB,
and FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced reads on gfx950 -> x2
(confirmed here on head_tail_fwd_kernel: 2 x 409 684 KiB = 839 MB = its 838.9 MB of input).
usage: pmc_traffic.py fetch.db write.db out.json"""
import json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from db_text_minimal_amd._lib import source_stamp  # (hashes csrc/: no GPU, no library load)

def load(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for k, avg, cnt in c.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? "
                                 "group by kernel_name", (counter, )):
        out[k.replace('(anonymous namespace)::', '').replace('void ', '')] = (avg, cnt)
    return out

f, w = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
res = {}
for k in sorted(set(f) & set(w)):
    name = k.split('(')[0].replace(', ', ',')
    rd, wr = 2.0 * f[k][0] * 1024, w[k][0] * 1024
    res[name] = {'launches_sampled': f[k][1], 'hbm_read_bytes_per_launch': round(rd), 'hbm_write_bytes_per_launch': round(wr),
                 'hbm_bytes_per_launch': round(rd + wr)}
json.dump({'csrc_stamp': source_stamp(),  # bench.py quotes these figures only while csrc/ still hashes to this
           'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1',
           'corrections': 'KiB -> bytes; FETCH_SIZE x2 (gfx950 wide-read under-count)', 'kernels': res}, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(res, indent=1)[:1500])
