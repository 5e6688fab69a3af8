#!/usr/bin/env python3
"""Sum a per-dispatch rocprofv3 counter (default SQ_INSTS_VALU) per kernel over a bench run: the instruction budget of a folding
step, to compare with what the part can issue.  usage: valu_budget.py <counter_collection.csv> <steps folded> [counter]"""
import collections
import csv
import sys

steps = float(sys.argv[2])   # rows folded by the run: warm-up + timed + the single-proof probe (count k_fold5<Fr> launches)
counter = sys.argv[3] if len(sys.argv) > 3 else "SQ_INSTS_VALU"
acc, calls = collections.Counter(), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != counter:
        continue
    name = r["Kernel_Name"].replace("void vz::", "").replace("vz::", "").split("(")[0]
    acc[name] += float(r["Counter_Value"])
    calls[name] += 1
setup = ("k_ckgen", "k_build_tables", "k_points_to_internal", "k_to_mont", "k_from_mont")
tot = sum(v for k, v in acc.items() if not k.startswith(setup))
print(f"{counter}: {tot / steps / 1e6:.1f} M per step over {steps:.0f} steps (set-up kernels excluded)")
for k, v in acc.most_common(24):
    print(f"  {k[:70]:70s} calls={calls[k]:6d}  {v / steps / 1e6:9.2f} M/step  {100 * v / tot:5.1f} %")
