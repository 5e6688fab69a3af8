import sys, time, numpy as np
sys.path.insert(0, '.')
from vimz_amd import hip, _lib
ctx = hip.Context(0)
rs = np.random.default_rng(1)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"      # plain | mult (tables of multiples: vimz_bases_precompute(7))
for n in (7700, 1536, 24000):
    B = ctx.bases_generate(_lib.CURVE_BN254_G1, n)
    if mode == "mult":
        B.precompute(7)
    dense = rs.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); dense[:, 3] &= np.uint64((1 << 60) - 1)
    v = ctx.vec_from_host(_lib.FIELD_BN254_FR, dense)
    ctx.msm_vec(B, v)
    ctx.set_profiling(True)
    ms = []
    for _ in range(20):
        t0 = time.time(); ctx.msm_vec(B, v); wall = (time.time() - t0) * 1e3
        ms.append((ctx.msm_last_profile()["ms"]["accumulate"], wall))
    ctx.set_profiling(False)
    ms.sort()
    print(mode, n, "kernel ms median %.3f min %.3f  wall median %.3f" % (ms[10][0], ms[0][0], sorted(w for _, w in ms)[10]))
