"""Fused single-launch MSM (k_msm_small) against the general pipeline around the size where one hands over to the other."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from vimz_amd import hip, _lib
ctx = hip.Context(0)
rs = np.random.default_rng(1)
for n in (16000, 24576, 27693, 32768, 40000, 49152):
    B = ctx.bases_generate(_lib.CURVE_BN254_G1, n)
    dense = rs.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); dense[:, 3] &= np.uint64((1 << 60) - 1)
    v = ctx.vec_from_host(_lib.FIELD_BN254_FR, dense)
    out = []
    for c in (0, 9, 11):
        ref = ctx.msm_vec(B, v, window_bits=c)
        ctx.set_profiling(True)
        ms = []
        for _ in range(10):
            t0 = time.time(); r = ctx.msm_vec(B, v, window_bits=c); wall = (time.time() - t0) * 1e3
            assert np.array_equal(np.asarray(r), np.asarray(ref))
            ms.append((sum(ctx.msm_last_profile()["ms"].values()), wall))
        ctx.set_profiling(False)
        ms.sort()
        out.append("c=%d gpu %.3f wall %.3f" % (c, ms[5][0], sorted(w for _, w in ms)[5]))
    print(n, " | ".join(out))
