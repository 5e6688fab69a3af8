#!/usr/bin/env python3
"""Stress: small (fused-kernel) MSMs on one context while other contexts keep the GPU busy with large MSMs.
Every result must equal the one computed on an idle GPU.  usage: stress_small_msm.py [seconds] [busy threads]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from vimz_amd import hip, _lib  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10
nbusy = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rs = np.random.default_rng(5)


def dense(n):
    a = rs.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a


big_host = dense(313000)
ctx = hip.Context(0)
B = ctx.bases_generate(_lib.CURVE_BN254_G1, 1 << 19)
small = [ctx.vec_from_host(_lib.FIELD_BN254_FR, dense(n)) for n in (7586, 7709, 1000, 16000)]
want = [ctx.msm_vec(B, v) for v in small]
_bv = ctx.vec_from_host(_lib.FIELD_BN254_FR, big_host)
big_want = ctx.msm_vec(B, _bv)
stop = False


big_res = []


def busy():
    c = hip.Context(0)
    v = c.vec_from_host(_lib.FIELD_BN254_FR, big_host)
    n = bad = 0
    while not stop:
        got = c.msm_vec(B, v)
        n += 1
        if not (np.asarray(got) == np.asarray(big_want)).all():
            bad += 1
    big_res.append((n, bad))
    v.free(); c.close()


def small_worker(res):
    c = hip.Context(0)
    vs = [c.vec_from_host(_lib.FIELD_BN254_FR, v.download()) for v in small]
    n = bad = 0
    while not stop:
        for v, w in zip(vs, want):
            got = c.msm_vec(B, v)
            n += 1
            if not (np.asarray(got) == np.asarray(w)).all():
                bad += 1
    res.append((n, bad))
    c.close()


th = [threading.Thread(target=busy) for _ in range(nbusy)]
res = []
th += [threading.Thread(target=small_worker, args=(res,)) for _ in range(3)]
for t in th:
    t.start()
time.sleep(secs)
stop = True
for t in th:
    t.join()
print("small MSMs run / wrong per worker:", res, " large:", big_res)
