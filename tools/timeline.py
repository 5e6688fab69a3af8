#!/usr/bin/env python3
"""Print a window of a rocprofv3 kernel trace as a timeline (start, duration, queue, kernel, grid) — to see what one folding
step's chain waits for.  usage: timeline.py <kernel_trace.csv> [start fraction 0..1] [window ms]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
win = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
t_lo = min(int(r["Start_Timestamp"]) for r in rows)
t_hi = max(int(r["End_Timestamp"]) for r in rows)
t0 = t_lo + int((t_hi - t_lo) * frac)
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t0 + win * 1e6]
sel.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in sel:
    name = r["Kernel_Name"].replace("void vz::", "").replace("vz::", "").split("(")[0].split("<")[0]
    tmpl = r["Kernel_Name"]
    fld = "Fq" if "Fp<vz::BnFq>" in tmpl and "Fp29" not in tmpl else ""
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q={r.get('Queue_Id', '?'):>3s}  {name:22s} grid={r['Grid_Size_X']:>8s} {fld}")
