#!/usr/bin/env python3
"""Latency of the hash-only Poseidon-chain pass (vimz_ivc_row_digests: k_wit_chains without wires) for a few row counts — one 16-lane
group per chain, so the time is one chain's latency whatever the count.  usage: digest_bench.py [transformation] [resolution]"""
import sys
import time

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import numpy as np  # noqa: E402
import bench  # noqa: E402
from vimz_amd import folding, hip  # noqa: E402


def main():
    t = sys.argv[1] if len(sys.argv) > 1 else "contrast"
    res = sys.argv[2] if len(sys.argv) > 2 else "HD"
    rows, z0 = bench.build_inputs(t, res)
    rows = np.stack(rows[:512])
    ctx = hip.Context(0)
    circuit, params = folding.prepare_folding(ctx, t, res)
    ivc = hip.IVC(ctx, circuit, params.ck, params.secondary_key(), max_batch=24)
    import os
    os.environ.setdefault("VIMZ_HEAD_ROWS", "24")
    for n in (97, 128, 256, 512):          # (above 96 rows the pass runs on the GPU; below, on the host pool)
        ivc.row_digests(rows[:n])
        ts = []
        for _ in range(5):
            t0 = time.time(); ivc.row_digests(rows[:n]); ts.append(time.time() - t0)
        print(f"{t} {res}: row digests of {n} rows: median {1e3 * sorted(ts)[2]:.2f} ms, min {1e3 * min(ts):.2f} ms")


if __name__ == "__main__":
    main()
