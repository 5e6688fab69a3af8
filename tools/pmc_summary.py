#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel (and per grid size for k_accum).
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and, on gfx950, FETCH_SIZE
counts 64 B per 128-B request for 16-B-per-lane reads (MI355X_MICROARCH.md §HBM) — calibrated in the same run on
k_points_to_internal (a pure 16 B/lane streaming read of 32 MiB reports 16.2 MiB).
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("void vz::", "").replace("vz::", "").split("(")[0]
        key = (name, int(r["Grid_Size"])) if name.startswith("k_accum") else (name, 0)
        acc[key].append(float(r["Counter_Value"]))
    return acc


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for key in sorted(f, key=lambda k: -sum(f[k])):
        name = key[0] + (f"[grid={key[1]}]" if key[1] else "")
        fk = sum(f[key]) / len(f[key])
        wk = sum(w.get(key, [0])) / max(1, len(w.get(key, [0])))
        out[name] = {"launches": len(f[key]), "fetch_kib_raw": fk, "write_kib_raw": wk, "hbm_bytes_per_launch": (2 * fk + wk) * 1024}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in list(out.items())[:14]:
        print(f"{k[:52]:52s} n={v['launches']:5d} fetch={v['fetch_kib_raw']:10.1f} KiB write={v['write_kib_raw']:10.1f} KiB -> {v['hbm_bytes_per_launch']/1e6:8.2f} MB/launch")


if __name__ == "__main__":
    main()
