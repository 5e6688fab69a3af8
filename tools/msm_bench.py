#!/usr/bin/env python3
"""Time the MSM pipeline at the shapes of one folding step: dense 254-bit scalars (MSM(T))
and witness-like scalars (MSM(W)).  Prints per-kernel HIP-event times.  GPU box only."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from vimz_amd import hip, _lib  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [315_000, 925_000]
    ctx = hip.Context(0)
    print(json.dumps(ctx.device_info()))
    rs = np.random.default_rng(1)
    for n in sizes:
        t0 = time.time()
        B = ctx.bases_generate(_lib.CURVE_BN254_G1, n)
        t_gen = time.time() - t0
        dense = rs.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        dense[:, 3] &= np.uint64((1 << 60) - 1)
        wit = np.zeros((n, 4), dtype=np.uint64)
        u = rs.random(n)
        wit[:, 0] = np.where(u < 0.8, rs.integers(0, 2, n), rs.integers(0, 256, n)).astype(np.uint64)
        full = u > 0.95
        wit[full] = dense[full]
        for name, sc in (("dense", dense), ("witness", wit)):
            v = ctx.vec_from_host(_lib.FIELD_BN254_FR, sc)
            for c in ([0, 12, 13, 14, 15] if name == "dense" else [0, 12, 13, 14]):
                ctx.set_profiling(False)
                ctx.msm_vec(B, v, window_bits=c)  # warm
                ctx.set_profiling(True)
                best = None
                for _ in range(3):
                    t0 = time.time()
                    ctx.msm_vec(B, v, window_bits=c)
                    wall = (time.time() - t0) * 1e3
                    p = ctx.msm_last_profile()
                    tot = sum(p["ms"].values())
                    if best is None or tot < best[0]:
                        best = (tot, wall, p)
                tot, wall, p = best
                print(f"n={n} {name:8s} c={p['window_bits']:2d} K={p['windows']:2d} subs={p['sub_buckets']:7d} entries={p['entries']:9d} "
                      f"gpu={tot:7.3f} ms wall={wall:7.3f} ms  " + " ".join(f"{k}={x:.3f}" for k, x in p["ms"].items()))
            v.free()
        # window tables with ONE shared bucket set (vimz_bases_precompute(c), c = 13..16): fewer digits per scalar, the 2^(c-1) buckets
        # reduced as virtual windows of 1024
        v = ctx.vec_from_host(_lib.FIELD_BN254_FR, dense)
        for c in (13, 14, 15, 16):
            B.precompute(c)
            ctx.set_profiling(False)
            ctx.msm_vec(B, v)
            ctx.set_profiling(True)
            best = None
            for _ in range(3):
                t0 = time.time()
                ctx.msm_vec(B, v)
                wall = (time.time() - t0) * 1e3
                p = ctx.msm_last_profile()
                tot = sum(p["ms"].values())
                if best is None or tot < best[0]:
                    best = (tot, wall, p)
            tot, wall, p = best
            print(f"n={n} dense    tables c={c:2d} K={p['windows']:2d} subs={p['sub_buckets']:7d} entries={p['entries']:9d} "
                  f"gpu={tot:7.3f} ms wall={wall:7.3f} ms  " + " ".join(f"{k}={x:.3f}" for k, x in p["ms"].items()))
        v.free()
        B.free()
        print(f"n={n}: ck generation {t_gen*1e3:.1f} ms")


if __name__ == "__main__":
    main()
