#!/usr/bin/env python3
"""VERDICT r5 #2a, a gated experiment: nineteen in twenty rows of a step circuit are b·(b − 1) = 0 (circuit/builder.hpp: group_boolean_rows_first), so on
those rows the cross term of the running instance (a_i = AZ_1[i], u_1) and a fresh one (b_i in {0, 1}) is  T_i = b_i ? a_i − u_1 : −a_i, and with
C_A = Σ_bool a_i·ck_i (kept by linearity: C_A += r·S_1) and S_1 = Σ_{b_i = 1} ck_i (a unit-scalar sum)
    comm_T|bool = 2·MSM_{b=1}(T) + u_1·S_1 − C_A = C_A + 2·MSM_{b=0}(T) − u_1·S_1:
the dense MSM only needs the SMALLER of the two sets.  This counts, over rows of the sample image through contrast_step_HD folded with 128-bit
challenges, how many points that leaves against the non-zero entries of T the dense MSM takes today.  Gate: build it if >= 20 % fewer.
usage: t_boolean_rows.py [rows = 256] [first row = 150]"""
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
    mats, csr = [], {}
    for m in "ABC":
        rp, col, coef = c.csr(m)
        csr[m] = (rp, col, coef)
        mats.append((np.repeat(np.arange(len(rp) - 1, dtype=np.uint32), np.diff(rp).astype(np.int64)), col, dict_canon[coef]))
    # the boolean rows: A = one wire with coefficient 1, B = that wire − 1, C empty
    rpA, colA, coefA = csr["A"]; rpB, colB, coefB = csr["B"]; rpC = csr["C"][0]
    lenA, lenB, lenC = np.diff(rpA), np.diff(rpB), np.diff(rpC)
    one = [0, 0, 0, 0]; one[0] = 1
    is_one = np.all(dict_canon == np.array(one, dtype=np.uint64), axis=1)
    cand = np.nonzero((lenA == 1) & (lenB == 2) & (lenC == 0))[0]
    wire = colA[rpA[cand]]
    ok = is_one[coefA[rpA[cand]]] & (wire != 0) & (colB[rpB[cand]] == 0) & (colB[rpB[cand] + 1] == wire) & is_one[coefB[rpB[cand] + 1]]
    brow, bwire = cand[ok], wire[ok]
    S = hip.R1CSShape(ctx, _lib.FIELD_BN254_FR, c.n_constraints, c.n_wires, *mats)
    ck = ctx.bases_generate(_lib.CURVE_BN254_G1, c.n_constraints)
    prover = hip.Prover(ctx, c, ck, max_batch=16)
    prover.reset(z0)
    q = _lib.MODULUS[0]
    rs = np.random.default_rng(9)
    z_run = ctx.vec_alloc(_lib.FIELD_BN254_FR, c.n_wires)
    u_run = 0
    stats = []
    for b0 in range(0, n_rows, 16):
        chunk = rows[b0:b0 + 16]
        wires, states, status = prover.witness(chunk)
        assert not np.any(status)
        prover.reset([sum(int(states[-1, i, k]) << (64 * k) for k in range(4)) for i in range(states.shape[1])])
        for k in range(len(chunk)):
            z2 = ctx.vec_from_host(_lib.FIELD_BN254_FR, wires[k])
            T, _ = S.commit_T(ck, z_run, u_run, z2, 1)
            t = np.ascontiguousarray(T.download())
            nzmask = np.any(t != 0, axis=1)
            bits = wires[k][bwire]
            assert np.all(bits[:, 1:] == 0) and np.all(bits[:, 0] <= 1), "a row taken for boolean is not"
            b1 = bits[:, 0] == 1
            nzb = nzmask[brow]
            n1, n0 = int(np.count_nonzero(nzb & b1)), int(np.count_nonzero(nzb & ~b1))
            other = int(np.count_nonzero(nzmask)) - n1 - n0
            stats.append({"step": b0 + k, "nonzero": int(np.count_nonzero(nzmask)), "bool_b1_nonzero": n1, "bool_b0_nonzero": n0, "other_nonzero": other,
                          "fresh_ones": int(np.count_nonzero(b1)), "dense_with_trick": other + min(n0, n1)})
            r = (int(rs.integers(0, 1 << 62)) | int(rs.integers(0, 1 << 62)) << 62 | 1 << 128) % q
            hip.vec_axpy(ctx, z_run, r, z2)
            u_run = (u_run + r) % q
            T.free(); z2.free()
    body = [s for s in stats if s["step"] >= 2]
    mean = lambda key: sum(s[key] for s in body) / max(1, len(body))
    out = {"config": "contrast_step_HD", "rows": n_rows, "first_row": first, "constraints": c.n_constraints, "boolean_rows": int(len(brow)),
           "mean_nonzero_per_step": mean("nonzero"), "mean_bool_b1_nonzero": mean("bool_b1_nonzero"), "mean_bool_b0_nonzero": mean("bool_b0_nonzero"),
           "mean_other_nonzero": mean("other_nonzero"), "mean_fresh_ones": mean("fresh_ones"), "mean_dense_with_trick": mean("dense_with_trick"),
           "dense_points_saved": 1.0 - mean("dense_with_trick") / max(1.0, mean("nonzero")), "unit_scalar_sum_points_added": mean("fresh_ones"),
           "a_late_step": stats[-1], "gate": "build it if the dense MSM loses >= 20 % of its points (VERDICT r5 #2a)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
