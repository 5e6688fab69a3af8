#!/usr/bin/env python3
"""Large-MSM time on every curve of the two cycles (BN254 G1 / Grumpkin / Pallas / Vesta): dense 254/255-bit scalars, n points,
HIP-event phases of the general pipeline.  GPU box only.  usage: msm_curves_bench.py [n ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from vimz_amd import hip, _lib  # noqa: E402

NAMES = {0: "bn254_g1", 1: "grumpkin", 2: "pallas", 3: "vesta"}


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [305_185, 1 << 20]
    ctx = hip.Context(0)
    rs = np.random.default_rng(1)
    for n in sizes:
        dense = rs.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        dense[:, 3] &= np.uint64((1 << 60) - 1)
        for cid in range(4):
            B = ctx.bases_generate(cid, n)
            v = ctx.vec_from_host(_lib.CURVE_SCALAR_FIELD[cid], dense)
            ctx.set_profiling(False)
            ctx.msm_vec(B, v)
            ctx.set_profiling(True)
            best = None
            for _ in range(5):
                t0 = time.time()
                ctx.msm_vec(B, v)
                wall = (time.time() - t0) * 1e3
                prof = ctx.msm_last_profile()
                tot = sum(prof["ms"].values())
                if best is None or tot < best[0]:
                    best = (tot, wall, prof)
            tot, wall, prof = best
            print(f"{NAMES[cid]:9s} n={n:8d} c={prof['window_bits']} K={prof['windows']} gpu={tot:7.3f} ms wall={wall:7.3f} ms  "
                  + " ".join(f"{k}={x:.3f}" for k, x in prof["ms"].items()), flush=True)
            v.free(); B.free()
    ctx.close()


if __name__ == "__main__":
    main()
