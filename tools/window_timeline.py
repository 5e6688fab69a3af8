#!/usr/bin/env python3
"""The timed pass of a bench.py run under `rocprofv3 --kernel-trace` as a timeline: every kernel between the two trace markers
(k_trace_marker, bench.py: around the LAST timed pass), start relative to marker 1, duration, queue, kernel name, grid.
usage: window_timeline.py <kernel_trace.csv> [first ms] [last ms] [min duration us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
lo_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
marks = sorted((int(r["Start_Timestamp"]), int(r["Grid_Size_X"]) // 64) for r in rows if "k_trace_marker" in r["Kernel_Name"])
m1 = [t for t, g in marks if g == 1]
m2 = [t for t, g in marks if g == 2]
if not m1 or not m2:
    sys.exit("no trace markers in the trace")
t0, t1 = m1[-1], m2[-1]
print(f"# timed pass: {(t1 - t0) / 1e6:.3f} ms between the markers")
sel = sorted((r for r in rows if t0 <= int(r["Start_Timestamp"]) <= t1), key=lambda r: int(r["Start_Timestamp"]))
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if not (lo_ms * 1e6 <= s - t0 <= hi_ms * 1e6) or (e - s) / 1e3 < min_us:
        continue
    name = r["Kernel_Name"].replace("void vz::", "").replace("vz::", "").split("(")[0].split("<")[0]
    fld = "Fq" if "Fp<vz::BnFq>" in r["Kernel_Name"] and "Fp29" not in r["Kernel_Name"] else ""
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q={r.get('Queue_Id', '?'):>3s}  {name:24s} grid={r['Grid_Size_X']:>8s} {fld}")
