#!/usr/bin/env python3
"""Split a rocprofv3 kernel trace CSV by (kernel, grid size): the same kernel runs at two sizes per step
(MSM(T) on the main stream, MSM(W) on the second one).  usage: split_kernel_trace.py <kernel_trace.csv> [name filter]"""
import collections
import csv
import sys

acc = collections.defaultdict(list)
flt = sys.argv[2] if len(sys.argv) > 2 else "k_accum"
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void vz::", "").replace("vz::", "").split("(")[0]
    if flt not in name:
        continue
    acc[(name, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{name:28s} grid={grid:8d} launches={len(v):5d} avg={sum(v)/len(v):9.1f} us  median={v[len(v)//2]:9.1f} us  min={v[0]:9.1f}  max={v[-1]:9.1f}")
