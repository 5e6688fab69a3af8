#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py: the large MSM's kernel sequence (k_hist_lds .. k_reduce around a k_accum of the MSM(T) grid) —
per kernel the mean execution time and the mean gap since the previous kernel of the same sequence ended: how much of the MSM's in-bench
latency is kernels running and how much is waiting between launches.  usage: msm_chain_gaps.py <kernel_trace.csv> [k_accum grid size]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
key = "Queue_Id" if "Queue_Id" in rows[0] else None
ev = []
for r in rows:
    name = r["Kernel_Name"].replace("void vz::", "").replace("vz::", "").split("(")[0].split("<")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), r.get(key, "") if key else "", r.get("Thread_Id", "")))
ev.sort()
if not grid:      # the most common k_accum grid among the larger ones = MSM(T)
    c = collections.Counter(e[3] for e in ev if e[2] == "k_accum")
    grid = max(c, key=lambda g: (c[g] > 20, g))
seq = ["k_fold_cross", "k_hist_lds", "k_block_prefix", "k_scan", "k_prefix_scan", "k_scatter_lds", "k_accum", "k_combine", "k_combine_heavy2", "k_reduce"]
by_q = collections.defaultdict(list)
for e in ev:
    by_q[e[4]].append(e)
exe = collections.defaultdict(list); gap = collections.defaultdict(list); total = []
for q, lst in by_q.items():
    for i, e in enumerate(lst):
        if e[2] != "k_accum" or e[3] != grid:
            continue
        # walk back to k_hist_lds and forward to k_reduce within this queue
        lo = i
        while lo > 0 and lst[lo][2] != "k_hist_lds":
            lo -= 1
        if lo > 0 and lst[lo - 1][2] == "k_fold_cross":
            lo -= 1
        hi = i
        while hi + 1 < len(lst) and lst[hi][2] != "k_reduce":
            hi += 1
        chain = lst[lo:hi + 1]
        names = [c[2] for c in chain]
        if names[-1] != "k_reduce" or "k_hist_lds" not in names or len(chain) > 12:
            continue
        for k, c in enumerate(chain):
            exe[c[2]].append(c[1] - c[0])
            if k:
                gap[c[2]].append(c[0] - chain[k - 1][1])
        total.append(chain[-1][1] - chain[0][0])
print(f"{len(total)} large MSMs (k_accum grid {grid}); mean span first kernel start -> k_reduce end: {sum(total) / max(1, len(total)) / 1e3:.1f} us")
se = sg = 0
for n in seq:
    if exe[n]:
        e_ = sum(exe[n]) / len(exe[n]) / 1e3
        g_ = sum(gap[n]) / len(gap[n]) / 1e3 if gap[n] else 0.0
        se += e_; sg += g_
        print(f"  {n:18s} runs {e_:7.1f} us   waits {g_:7.1f} us after its predecessor")
print(f"  sum of kernels {se:.1f} us, sum of gaps {sg:.1f} us")
