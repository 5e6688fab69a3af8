#!/usr/bin/env python3
"""Times the hash-only IVC state chain (vimz_ivc_state_chain) alone: what a later row segment's start state costs.
usage: state_chain_bench.py [transformation] [resolution]"""
import sys
import time

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402
from vimz_amd import folding, hip  # noqa: E402

t, res = (sys.argv[1] if len(sys.argv) > 1 else "contrast"), (sys.argv[2] if len(sys.argv) > 2 else "HD")
ctx = hip.Context(0)
circuit, params = folding.prepare_folding(ctx, t, res)
rows, z0 = bench.build_inputs(t, res)
ivc = hip.IVC(ctx, circuit, params.ck, params.secondary_key(), max_batch=8)
for n in (16, 96, 240, 720, len(rows)):
    n = min(n, len(rows))
    ivc.state_chain(z0, rows[:n])
    ts = []
    for _ in range(5):
        t0 = time.time(); ivc.state_chain(z0, rows[:n]); ts.append(time.time() - t0)
    print(f"{t} {res}: state chain over {n} rows: median {sorted(ts)[2] * 1e3:.2f} ms ({sorted(ts)[2] / n * 1e6:.1f} us per row)")
