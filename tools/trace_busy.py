#!/usr/bin/env python3
"""GPU occupancy over time from a rocprofv3 kernel trace: share of the window with at least one kernel running, mean number of
kernels running side by side, and the idle gaps.  The window is bench.py's TIMED REGION when the trace holds its markers (the empty
kernel k_trace_marker with 1 workgroup before and 2 workgroups after it: vimz_trace_marker) — the fold itself, not set-up, compression
or the extras —, else the given fractions of the whole trace.
usage: trace_busy.py <kernel_trace.csv> [from fraction] [to fraction]"""
import csv
import sys

raw = list(csv.DictReader(open(sys.argv[1])))
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in raw]


def grid(r):
    for k in ("Grid_Size_X", "Grid_Size", "Workgroup_Count_X"):
        if k in r and r[k] not in ("", None):
            return int(r[k])
    return 0


marks = sorted((int(r["End_Timestamp"]), int(r["Start_Timestamp"]), grid(r)) for r in raw if "k_trace_marker" in r["Kernel_Name"])
lo = min(r[0] for r in rows); hi = max(r[1] for r in rows)
a = b = None
if len(marks) >= 2:
    # grid sizes are reported in work-items (64 per workgroup) or in workgroups, depending on the profiler's version: the smaller is marker 1
    g1 = min(m[2] for m in marks)
    first = [m for m in marks if m[2] == g1]
    second = [m for m in marks if m[2] != g1]
    if first and second:
        a = first[0][0]                                    # end of marker 1
        b = [m[1] for m in second if m[1] > a][0]          # start of the marker 2 that follows it
        print(f"window: bench.py's timed region (between trace markers), {(b - a) / 1e6:.2f} ms")
if a is None:
    f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.55
    f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
    a, b = lo + (hi - lo) * f0, lo + (hi - lo) * f1
    print(f"window: {f0:.2f} .. {f1:.2f} of the whole trace (no markers found)")
ev = []
per_kernel = {}
for s, e, name in rows:
    if "k_trace_marker" in name:
        continue
    s, e = max(s, a), min(e, b)
    if e > s:
        ev.append((s, 1)); ev.append((e, -1))
        k = name.split("(")[0].split("<")[0]
        per_kernel[k] = per_kernel.get(k, 0) + (e - s)
ev.sort()
busy = 0; area = 0; cur = 0; last = a; gaps = []
for t, d in ev:
    if cur > 0:
        busy += t - last; area += cur * (t - last)
    elif t > last:
        gaps.append(t - last)
    cur += d; last = t
if b > last:
    gaps.append(b - last)
span = b - a
print(f"window {span / 1e6:.1f} ms: busy {100 * busy / span:.1f} %, mean kernels in flight {area / span:.2f}")
gaps.sort(reverse=True)
print("idle gaps: total %.2f ms, count %d, largest (us): %s" % (sum(gaps) / 1e6, len(gaps), [round(g / 1e3, 1) for g in gaps[:8]]))
top = sorted(per_kernel.items(), key=lambda kv: -kv[1])[:8]
print("kernel time inside the window (sum over launches, % of the window): " + ", ".join(f"{k.split('::')[-1]} {100 * v / span:.0f}" for k, v in top))
