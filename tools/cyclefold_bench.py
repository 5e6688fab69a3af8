#!/usr/bin/env python3
"""Nova + CycleFold IVC (vimz_cf_*, the reference's Sonobe backend): steps/s over rows of the sample image, with the phase split.
usage: cyclefold_bench.py [transformation] [resolution] [rows] [batch]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from vimz_amd import hip, folding  # noqa: E402
from bench import build_inputs  # noqa: E402


def main():
    op = sys.argv[1] if len(sys.argv) > 1 else "contrast"
    res = sys.argv[2] if len(sys.argv) > 2 else "HD"
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    ctx = hip.Context(0)
    circuit, params = folding.prepare_folding(ctx, op, res, backend="sonobe")
    steps, z0 = build_inputs(op, res)
    steps = np.stack(steps[:rows])
    z0 = [int(x) for x in z0]
    cf = hip.CycleFoldIVC(ctx, circuit, params.ck, params.secondary_key(), max_batch=batch or folding.default_batch(circuit))
    cf.reset(z0); cf.fold(steps[:16])          # warm-up
    assert cf.verify(16, z0) == 0
    cf.reset(z0)
    t0 = time.time(); cf.fold(steps[:rows]); dt = time.time() - t0
    t1 = time.time(); ok = cf.verify(rows, z0); tv = time.time() - t1
    prof = cf.profile()
    print(json.dumps({"metric": "cyclefold_folding_steps_per_sec", "workload": f"{op}_step_{res}", "value": rows / dt, "rows": rows, "verified": ok == 0, "verify_s": tv,
                      "info": cf.info(), "ms_per_step": {k: 1e3 * s / max(1, rows) for k, (s, n) in prof.items()}}))
    cf.close(); params.free(); ctx.close()


if __name__ == "__main__":
    main()
