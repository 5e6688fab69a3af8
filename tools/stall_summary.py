#!/usr/bin/env python3
"""Per-kernel means of SQ counters from a rocprofv3 --pmc counter_collection.csv: where a kernel's wave cycles go — parked on s_waitcnt / barriers
(SQ_WAIT_ANY), stalled at issue (SQ_WAIT_INST_ANY), issuing (SQ_ACTIVE_INST_ANY); MI355X_MICROARCH.md: the three are disjoint and sum to about
SQ_WAVE_CYCLES.  usage: stall_summary.py <counter_collection.csv> [kernel substring, default k_accum]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else "k_accum"
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in rows:
    name = r["Kernel_Name"]
    if want not in name:
        continue
    key = (name.split("(")[0].split("<")[0].replace("void vz::", ""), r.get("Grid_Size", r.get("Grid_Size_X", "?")))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[key][r["Counter_Name"]] += 1
for key in sorted(acc, key=lambda k: -sum(cnt[k].values())):
    c = {k: acc[key][k] / cnt[key][k] for k in acc[key]}
    n = max(cnt[key].values())
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    line = f"{key[0]} grid={key[1]} launches={n}: " + ", ".join(f"{k}={v:.3g}" for k, v in sorted(c.items()))
    if wc:
        line += " | share of wave cycles: " + ", ".join(f"{k.replace('SQ_', '')} {100 * c[k] / wc:.1f} %" for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in c)
    print(line)
