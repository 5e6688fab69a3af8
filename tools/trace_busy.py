#!/usr/bin/env python3
"""GPU occupancy over time from a rocprofv3 kernel trace: share of the window with at least one kernel running, mean number of
kernels running side by side, and the idle gaps.  usage: trace_busy.py <kernel_trace.csv> [from fraction] [to fraction]"""
import csv
import sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
lo = min(r[0] for r in rows); hi = max(r[1] for r in rows)
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.55
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
a, b = lo + (hi - lo) * f0, lo + (hi - lo) * f1
ev = []
for s, e, _ in rows:
    s, e = max(s, a), min(e, b)
    if e > s:
        ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; area = 0; cur = 0; last = a; gaps = []
for t, d in ev:
    if cur > 0:
        busy += t - last; area += cur * (t - last)
    elif t > last:
        gaps.append(t - last)
    cur += d; last = t
span = b - a
print(f"window {span / 1e6:.1f} ms: busy {100 * busy / span:.1f} %, mean kernels in flight {area / span:.2f}")
gaps.sort(reverse=True)
print("idle gaps: total %.2f ms, count %d, largest (us): %s" % (sum(gaps) / 1e6, len(gaps), [round(g / 1e3, 1) for g in gaps[:8]]))
