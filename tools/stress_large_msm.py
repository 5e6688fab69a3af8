#!/usr/bin/env python3
"""Stress of the large MSM's tails (k_combine's in-kernel stage 2 of the split buckets, k_reduce_planes' last-workgroup fold): the same MSM repeated on
several contexts side by side, over data that makes buckets ordinary, heavy and very heavy — every repetition must give the same point, and the table
path (one bucket set, bit planes) the same point as the path without tables.  usage: stress_large_msm.py [n] [repetitions] [contexts]"""
import sys
import threading

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vimz_amd import _lib, hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 305185
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
nctx = int(sys.argv[3]) if len(sys.argv) > 3 else 3
r = _lib.MODULUS[0]
rs = np.random.default_rng(3)


def limbs(vals):
    a = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(4):
            a[i, k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return a


def dense(m):
    a = rs.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a


datasets = {}
datasets["dense"] = dense(n)
w = np.zeros((n, 4), dtype=np.uint64)                      # witness-like: bits, bytes, a few full-width values
w[:, 0] = rs.integers(0, 2, size=n, dtype=np.uint64)
idx = rs.choice(n, n // 8, replace=False); w[idx, 0] = rs.integers(2, 256, size=len(idx), dtype=np.uint64)
idx = rs.choice(n, n // 20, replace=False); w[idx] = dense(len(idx))
datasets["witness"] = w
h = dense(n)                                                # two values repeated: very heavy buckets in every window
h[: n // 2] = h[0]; h[n // 2: n - n // 8] = h[n // 2]
datasets["two_values"] = h
m1 = limbs([r - 1]); h2 = np.repeat(m1, n, axis=0); h2[::7] = dense(len(h2[::7]))
datasets["minus_one"] = h2

bad = []


def worker(k):
    ctx = hip.Context(0)
    plain = ctx.bases_generate(_lib.CURVE_BN254_G1, n, b"stress")
    tab = ctx.bases_generate(_lib.CURVE_BN254_G1, n, b"stress").precompute(15)
    for name, data in datasets.items():
        for split in (False, True):
            vec = ctx.vec_from_host(_lib.FIELD_BN254_FR, data)
            ref = ctx.msm_vec(plain, vec, split_ones=split).tolist()
            for i in range(reps):
                got = ctx.msm_vec(tab, vec, split_ones=split).tolist()
                if got != ref:
                    bad.append((k, name, split, i))
            vec.free()
    plain.free(); tab.free(); ctx.close()


ts = [threading.Thread(target=worker, args=(k,)) for k in range(nctx)]
for t in ts:
    t.start()
for t in ts:
    t.join()
print(f"{nctx} contexts x {len(datasets)} data sets x 2 x {reps} table MSMs of {n} points: {len(bad)} differ from the MSM without tables", bad[:10])
sys.exit(1 if bad else 0)
