#!/usr/bin/env python3
"""VERDICT r4 #9, a gated experiment: how often do the non-zero entries of a REAL cross-term vector T repeat?  If many of them sat in groups of equal
scalars, the large MSM could first add the bases of each group (one addition per point) and run Pippenger over the group sums only.
Rows of the sample image through the contrast_step_HD circuit (GPU witness program), folded with 128-bit challenges through the R1CS seam
(vimz_commit_T); for every step the cross term T of the running and the fresh instance is downloaded and its values are grouped.
usage: t_multiplicity.py [rows = 256] [first row = 150]"""
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402
from vimz_amd import _lib, hip  # noqa: E402
from vimz_amd.circuit import Circuit  # noqa: E402


def main():
    n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    rows, z0 = bench.build_inputs("contrast", "HD")
    rows = np.ascontiguousarray(rows[[(first + i) % len(rows) for i in range(n_rows)]])
    c = Circuit.for_resolution("contrast", "HD")
    ctx = hip.Context(0)
    dict_canon = c.export("DICT_CANON", np.uint64).reshape(-1, 4)
    mats = []
    for m in "ABC":
        rp, col, coef = c.csr(m)
        mats.append((np.repeat(np.arange(len(rp) - 1, dtype=np.uint32), np.diff(rp).astype(np.int64)), col, dict_canon[coef]))
    S = hip.R1CSShape(ctx, _lib.FIELD_BN254_FR, c.n_constraints, c.n_wires, *mats)
    ck = ctx.bases_generate(_lib.CURVE_BN254_G1, c.n_constraints)
    prover = hip.Prover(ctx, c, ck, max_batch=16)
    prover.reset(z0)
    q = _lib.MODULUS[0]
    rs = np.random.default_rng(9)
    z_run = ctx.vec_alloc(_lib.FIELD_BN254_FR, c.n_wires)
    u_run = 0
    stats = []
    zs = [int(x) for x in z0]
    for b0 in range(0, n_rows, 16):
        chunk = rows[b0:b0 + 16]
        wires, states, status = prover.witness(chunk)          # (n, n_wires, 4) canonical, from the prover's current state
        assert not np.any(status)
        prover.reset([sum(int(states[-1, i, k]) << (64 * k) for k in range(4)) for i in range(states.shape[1])])      # the next chunk continues the chain
        for k in range(len(chunk)):
            z2 = ctx.vec_from_host(_lib.FIELD_BN254_FR, wires[k])
            T, _ = S.commit_T(ck, z_run, u_run, z2, 1)
            t = np.ascontiguousarray(T.download())
            nz = t[np.any(t != 0, axis=1)]
            if len(nz):
                keys = nz.view(np.dtype((np.void, 32))).reshape(-1)
                _, counts = np.unique(keys, return_counts=True)
                in_groups = lambda g: int(counts[counts >= g].sum())
                stats.append({"step": b0 + k, "nonzero": int(len(nz)), "distinct": int(len(counts)), "in_groups_ge2": in_groups(2), "in_groups_ge8": in_groups(8),
                              "in_groups_ge64": in_groups(64), "largest_group": int(counts.max())})
            else:
                stats.append({"step": b0 + k, "nonzero": 0, "distinct": 0, "in_groups_ge2": 0, "in_groups_ge8": 0, "in_groups_ge64": 0, "largest_group": 0})
            r = (int(rs.integers(0, 1 << 62)) | int(rs.integers(0, 1 << 62)) << 62 | 1 << 128) % q
            hip.vec_axpy(ctx, z_run, r, z2)
            u_run = (u_run + r) % q
            T.free(); z2.free()
    body = [s for s in stats if s["step"] >= 2]          # (the first steps fold into an almost empty running instance)
    tot = sum(s["nonzero"] for s in body)
    out = {"config": "contrast_step_HD", "rows": n_rows, "first_row": first, "constraints": c.n_constraints,
           "mean_nonzero_per_step": tot / max(1, len(body)), "mean_distinct_per_step": sum(s["distinct"] for s in body) / max(1, len(body)),
           "share_of_nonzero_in_groups_ge2": sum(s["in_groups_ge2"] for s in body) / max(1, tot),
           "share_of_nonzero_in_groups_ge8": sum(s["in_groups_ge8"] for s in body) / max(1, tot),
           "share_of_nonzero_in_groups_ge64": sum(s["in_groups_ge64"] for s in body) / max(1, tot),
           "largest_group_max": max(s["largest_group"] for s in body), "first_steps": stats[:3], "a_late_step": stats[-1],
           "gate": "pre-summing the bases of equal scalars pays if >= 15 % of the non-zero entries sit in groups of >= 8 (VERDICT r4 #9)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
