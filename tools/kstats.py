#!/usr/bin/env python3
"""Pretty-print a rocprofv3 *_kernel_stats.csv (name, calls, avg us, total ms, %)."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    name = r["Name"].replace("void vz::", "").replace("vz::", "")
    print(name[:44].ljust(44), r["Calls"].rjust(6), "%10.1f us" % (float(r["AverageNs"]) / 1e3), "%9.1f ms" % (float(r["TotalDurationNs"]) / 1e6), r["Percentage"])
