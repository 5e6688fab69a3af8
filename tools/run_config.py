#!/usr/bin/env python3
"""Fold a few steps of a BASELINE.json configuration on the GPU, verify, and print rates (GPU box only).
usage: run_config.py <transformation> <resolution> [steps]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402
from vimz_amd import folding, hip  # noqa: E402


def main():
    t, res = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    ctx = hip.Context(0)
    t0 = time.time()
    circuit, params = folding.prepare_folding(ctx, t, res)
    t_key = time.time() - t0
    rows, z0 = bench.build_inputs(t, res)
    rows = rows[:steps + 8]
    P = hip.Prover(ctx, circuit, params.ck, max_batch=32)
    P.reset(z0)
    P.fold(rows[:8])
    ctx.sync()
    t0 = time.time()
    P.fold(rows[8:])
    ctx.sync()
    dt = time.time() - t0
    ok = P.verify() == 0
    print(json.dumps({"config": f"{t}_step_{res}", "constraints": circuit.n_constraints, "wires": circuit.n_wires, "steps": len(rows) - 8,
                      "steps_per_s": (len(rows) - 8) / dt, "ms_per_step": 1e3 * dt / (len(rows) - 8), "verified": ok, "setup_s": t_key,
                      "full_image_steps": folding.iteration_count(t, res), "full_image_fold_s_est": folding.iteration_count(t, res) * dt / (len(rows) - 8)}))
    P.close(); params.ck.free(); ctx.close()


if __name__ == "__main__":
    main()
