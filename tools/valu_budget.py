#!/usr/bin/env python3
"""Sum a per-dispatch rocprofv3 counter (default SQ_INSTS_VALU) per kernel over a bench run: the instruction budget of a folding
step, to compare with what the part can issue.  usage: valu_budget.py <counter_collection.csv> <steps folded> [counter] [split]
(split: the MSM kernels are listed per grid size too — the large MSM(T), the witness MSM(W) and the merge's launches apart)"""
import collections
import csv
import sys

steps = float(sys.argv[2])   # rows folded by the run: warm-up + timed + the single-proof probe (count k_fold5<Fr> launches)
counter = sys.argv[3] if len(sys.argv) > 3 else "SQ_INSTS_VALU"
split = len(sys.argv) > 4 and sys.argv[4] == "split"
acc, calls = collections.Counter(), collections.Counter()
gacc, gcalls = collections.Counter(), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != counter:
        continue
    name = r["Kernel_Name"].replace("void vz::", "").replace("vz::", "").split("(")[0]
    acc[name] += float(r["Counter_Value"])
    calls[name] += 1
    if split and name.startswith(("k_accum", "k_combine", "k_reduce", "k_msm_small", "k_hist_lds", "k_scatter_lds", "k_ones_partial")):
        g = (name, r.get("Grid_Size", "?"))
        gacc[g] += float(r["Counter_Value"]); gcalls[g] += 1
setup = ("k_ckgen", "k_build_tables", "k_points_to_internal", "k_to_mont", "k_from_mont")
tot = sum(v for k, v in acc.items() if not k.startswith(setup))
print(f"{counter}: {tot / steps / 1e6:.1f} M per step over {steps:.0f} steps (set-up kernels excluded)")
for k, v in acc.most_common(24):
    print(f"  {k[:70]:70s} calls={calls[k]:6d}  {v / steps / 1e6:9.2f} M/step  {100 * v / tot:5.1f} %")
if split:
    print("per grid size:")
    for (k, g), v in sorted(gacc.items(), key=lambda kv: -kv[1])[:30]:
        print(f"  {k[:50]:50s} grid={g:>9s} calls={gcalls[(k, g)]:6d}  {v / steps / 1e6:9.2f} M/step  {v / gcalls[(k, g)] / 1e6:8.3f} M/launch")
