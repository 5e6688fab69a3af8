"""Independent verifier of a Nova + CycleFold IVC proof (vimz_cf_*, vimz_amd/csrc/aug/cyclefold.hpp): Sonobe's Nova::verify restated
with the CPU oracle — the oracle's Poseidon hash (oracle/nova.hpp: nova_hash over Fr), its generic relaxed-R1CS check and its MSM over the
exported shapes, instances and witness vectors — plus the in-circuit folding relation of one step in Python integers / oracle curve
arithmetic.  Test infrastructure only: nothing of the product imports this."""
import hashlib
import struct

import numpy as np

from tests._oracle import from_limbs

CF_IO = 7


def limbs64(v):
    return [(int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]


def shape_digest(cf):
    """SHA3-256 of both shapes as CfMainCircuit::finish serialises them, truncated to 250 bits."""
    from vimz_amd import hip
    ex = lambda side, code: hip._export(cf.ctx.lib.vimz_cf_export, cf.h, side, code)
    info0, info1 = ex(0, hip.IX_INFO).view(np.uint64), ex(1, hip.IX_INFO).view(np.uint64)
    h = hashlib.sha3_256()
    h.update(struct.pack("<6Q", 0x31306d6663, int(info0[0]), int(info0[1]), cf.circuit.len_z, int(info0[2]), int(info1[0])))
    for side in (0, 1):
        for code in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9):
            a = ex(side, code)
            elem = 32 if code == 9 else 4
            h.update(struct.pack("<Q", len(a) // elem))
            h.update(a.tobytes())
    return int.from_bytes(h.digest(), "little") & ((1 << 250) - 1)


def hash_main(orc, dg, i, z0, z, U):
    """U = (W.x, W.y, E.x, E.y, u, x0, x1): H(H(dg, i, z0, z), u, x0, x1, limbs of W, limbs of E)"""
    st = orc.nova_hash(0, [dg, i] + list(z0) + list(z))
    return orc.nova_hash(0, [st, U[4], U[5], U[6]] + limbs64(U[0]) + limbs64(U[1]) + limbs64(U[2]) + limbs64(U[3]))


def hash_cf(orc, dg, cfU):
    """cfU = (W.x, W.y, E.x, E.y, u, x[0..7))"""
    lim = []
    for x in cfU[5:5 + CF_IO]:
        lim += limbs64(x)
    return orc.nova_hash(0, [dg, cfU[4]] + lim + list(cfU[0:4]))


def chal(h):
    return (1 << 128) + (int(h) & ((1 << 128) - 1))


def verify(orc, cf, ck1, ck2, n_steps, z0, check_commitments=True):
    """Nova::verify of Sonobe's IVC proof, restated; returns (failed checks, final state)."""
    from vimz_amd import hip
    failed = []
    info = cf.info()
    len_z = info["len_z"]
    par = from_limbs(cf.export(0, hip.IX_PARAMS))
    z0_x, z_n = par[1:1 + len_z], par[1 + len_z:]
    if z0_x != list(z0): failed.append("z0")
    if info["steps"] != n_steps: failed.append("step count")
    dg = shape_digest(cf)
    if dg != par[0]: failed.append("shape digest")
    U, cfU = from_limbs(cf.export(0, hip.IX_INSTANCE)), from_limbs(cf.export(1, hip.IX_INSTANCE))
    u = from_limbs(cf.export(0, hip.IX_FRESH_INSTANCE))
    if hash_main(orc, dg, n_steps, list(z0), z_n, U) != u[2]: failed.append("hash of the main running instance")
    if hash_cf(orc, dg, cfU) != u[3]: failed.append("hash of the CycleFold running instance")
    # running main pair
    tabs = cf.r1cs(0)
    Z, E = cf.export(0, hip.IX_RUNNING_Z), cf.export(0, hip.IX_RUNNING_E)
    nw = len(Z)
    if from_limbs(Z[0:1])[0] != U[4] or from_limbs(Z[-2:]) != U[5:7]: failed.append("main: instance scalars")
    if orc.r1cs_check_relaxed(0, tabs, nw, Z, u=U[4], E=E) != -1: failed.append("main: relaxed relation")
    # the last instance of F'
    z = cf.export(0, hip.IX_FRESH_Z)
    if from_limbs(z[0:1])[0] != 1 or from_limbs(z[-2:]) != u[2:4]: failed.append("fresh: instance scalars")
    if orc.r1cs_check_relaxed(0, tabs, nw, z) != -1: failed.append("fresh: relation")
    if check_commitments:
        bases = ck1.download(0, max(nw - 3, len(E)))
        if orc.msm(0, bases[:nw - 3], Z[1:nw - 2]) != (U[0], U[1]): failed.append("main: comm_W")
        if orc.msm(0, bases[:len(E)], E) != (U[2], U[3]): failed.append("main: comm_E")
        if orc.msm(0, bases[:nw - 3], z[1:nw - 2]) != (u[0], u[1]): failed.append("fresh: comm_W")
    # running CycleFold pair (over Fq, committed on Grumpkin)
    tabs2 = cf.r1cs(1)
    Z2, E2 = cf.export(1, hip.IX_RUNNING_Z), cf.export(1, hip.IX_RUNNING_E)
    n2 = len(Z2)
    if from_limbs(Z2[0:1])[0] != cfU[4] or from_limbs(Z2[-CF_IO:]) != cfU[5:5 + CF_IO]: failed.append("cyclefold: instance scalars")
    if orc.r1cs_check_relaxed(1, tabs2, n2, Z2, u=cfU[4], E=E2) != -1: failed.append("cyclefold: relaxed relation")
    if check_commitments:
        bases2 = ck2.download(0, max(n2 - 1 - CF_IO, len(E2)))
        if orc.msm(1, bases2[:n2 - 1 - CF_IO], Z2[1:n2 - CF_IO]) != (cfU[0], cfU[1]): failed.append("cyclefold: comm_W")
        if orc.msm(1, bases2[:len(E2)], E2) != (cfU[2], cfU[3]): failed.append("cyclefold: comm_E")
    return failed, z_n


def cyclefold_relation(orc, x):
    """The CycleFold circuit's statement for one instance's public elements x = (r, P1, P2, P3): P3 == P1 + r·P2 on BN254 G1."""
    r, p1, p2, p3 = x[0], (x[1], x[2]), (x[3], x[4]), (x[5], x[6])
    return orc.curve_add(0, p1, orc.curve_mul(0, p2, r)) == p3


# ---- merged proofs (vimz_cf_merge): replay of the segment records with hashlib, Python integers and the oracle's curve arithmetic -------------
MERGED_MAGIC = 0x32474d46435a56


def parse_merged_records(words, len_z):
    w = [int(x) for x in words]
    magic, S, lz, nw1, nc1, nw2, nc2, R = w[:8]
    assert magic == MERGED_MAGIC and lz == len_z and 1 <= R <= S
    run_start = w[8:8 + R]
    assert run_start[0] == 0 and all(a < b for a, b in zip(run_start, run_start[1:])) and run_start[-1] < S
    pos = 8 + R

    def el():
        nonlocal pos
        v = sum(w[pos + k] << (64 * k) for k in range(4))
        pos += 4
        return v
    segs = []
    for _ in range(S):
        s = {"n": w[pos]}
        pos += 1
        s["zs"] = [el() for _ in range(lz)]
        s["ze"] = [el() for _ in range(lz)]
        s["U"] = [el() for _ in range(7)]
        s["u"] = [el() for _ in range(4)]
        s["cfU"] = [el() for _ in range(5 + CF_IO)]
        s["T1"], s["T2"], s["Tc"] = (el(), el()), (el(), el()), (el(), el())
        segs.append(s)
    junctions = [((el(), el()), (el(), el())) for _ in range(R - 1)]
    assert pos == len(w)
    return {"shape": (nw1, nc1, nw2, nc2), "segs": segs, "run_start": run_start, "junctions": junctions}


def _b32(x):
    return int(x).to_bytes(32, "little")


def replay_merged(orc, rec, dg, len_z):
    """Returns (failed checks, accumulator dict: n, zs, ze, P = (cW, cE, u, x0, x1), Q = (qW, qE, qu, qx[7]))."""
    import hashlib
    sha = lambda b: hashlib.sha3_256(b).digest()
    chal = lambda h, tag, extra: int.from_bytes(sha(h + tag + extra)[:16], "little")
    pr, pq = orc.modulus[0], orc.modulus[1]
    axpy = lambda cid, a, r, b: orc.curve_add(cid, a, orc.curve_mul(cid, b, r))
    failed = []
    runs = rec.get("run_start", [0])
    bounds = list(zip(runs, runs[1:] + [len(rec["segs"])]))
    total = None
    for k, (lo, hi) in enumerate(bounds):
        f, acc = _replay_run(orc, rec["segs"][lo:hi], dg, len_z, sha, chal, axpy, pr, pq, lo)
        failed += f
        if total is None:
            total = acc
            continue
        Tp, Tq = rec["junctions"][k - 1]
        if total["ze"] != acc["zs"]: failed.append(f"run {k}: not adjacent")
        h = sha(b"vimz-cf-merge-node-v1" + total["h"] + acc["h"] + _b32(Tp[0]) + _b32(Tp[1]) + _b32(Tq[0]) + _b32(Tq[1]))
        rp, rq = chal(h, b"p", b""), chal(h, b"q", b"")
        P, Q, Pb, Qb = total["P"], total["Q"], acc["P"], acc["Q"]
        P[0] = axpy(0, P[0], rp, Pb[0]); P[1] = axpy(0, P[1], rp, axpy(0, Tp, rp, Pb[1]))
        P[2], P[3], P[4] = (P[2] + rp * Pb[2]) % pr, (P[3] + rp * Pb[3]) % pr, (P[4] + rp * Pb[4]) % pr
        Q[0] = axpy(1, Q[0], rq, Qb[0]); Q[1] = axpy(1, Q[1], rq, axpy(1, Tq, rq, Qb[1]))
        Q[2] = (Q[2] + rq * Qb[2]) % pq
        Q[3] = [(a + rq * b) % pq for a, b in zip(Q[3], Qb[3])]
        total["n"] += acc["n"]; total["ze"] = acc["ze"]; total["h"] = h
    return failed, total


def _replay_run(orc, segs, dg, len_z, sha, chal, axpy, pr, pq, first_index):
    failed = []
    h_prev = sha(b"vimz-cf-merge-v1" + _b32(dg) + len_z.to_bytes(8, "little"))
    acc = None
    for j, s in enumerate(segs, start=first_index):
        U, u, cfU = s["U"], s["u"], s["cfU"]
        if s["n"] == 0: failed.append(f"segment {j}: empty")
        if hash_main(orc, dg, s["n"], s["zs"], s["ze"], U) != u[2]: failed.append(f"segment {j}: hash of the main running instance")
        if hash_cf(orc, dg, cfU) != u[3]: failed.append(f"segment {j}: hash of the CycleFold running instance")
        if acc is not None and acc["ze"] != s["zs"]: failed.append(f"segment {j}: not adjacent")
        m = h_prev + s["n"].to_bytes(8, "little") + b"".join(_b32(x) for x in s["zs"] + s["ze"] + U + u + cfU)
        h = sha(m)
        t1, t2, tc = _b32(s["T1"][0]) + _b32(s["T1"][1]), _b32(s["T2"][0]) + _b32(s["T2"][1]), _b32(s["Tc"][0]) + _b32(s["Tc"][1])
        r1, r2, rc = chal(h, b"a", t1), chal(h, b"b", t1 + t2), chal(h, b"c", tc)
        h_prev = sha(h + b"n" + t1 + t2 + tc)
        if acc is None:
            acc = {"n": 0, "zs": s["zs"], "P": [(U[0], U[1]), (U[2], U[3]), U[4], U[5], U[6]], "Q": [(cfU[0], cfU[1]), (cfU[2], cfU[3]), cfU[4] % pq, [x % pq for x in cfU[5:]]]}
        else:
            P, Q = acc["P"], acc["Q"]
            P[0] = axpy(0, P[0], r1, (U[0], U[1]))
            P[1] = axpy(0, P[1], r1, axpy(0, s["T1"], r1, (U[2], U[3])))
            P[2], P[3], P[4] = (P[2] + r1 * U[4]) % pr, (P[3] + r1 * U[5]) % pr, (P[4] + r1 * U[6]) % pr
            Q[0] = axpy(1, Q[0], rc, (cfU[0], cfU[1]))
            Q[1] = axpy(1, Q[1], rc, axpy(1, s["Tc"], rc, (cfU[2], cfU[3])))
            Q[2] = (Q[2] + rc * cfU[4]) % pq
            Q[3] = [(a + rc * b) % pq for a, b in zip(Q[3], cfU[5:])]
        P = acc["P"]
        P[0] = axpy(0, P[0], r2, (u[0], u[1]))
        P[1] = axpy(0, P[1], r2, s["T2"])
        P[2], P[3], P[4] = (P[2] + r2) % pr, (P[3] + r2 * u[2]) % pr, (P[4] + r2 * u[3]) % pr
        acc["n"] += s["n"]
        acc["ze"] = s["ze"]
        acc["h"] = h_prev
    return failed, acc


def verify_merged(orc, merged, cf, ck1, ck2, num_steps, z0, check_commitments=True):
    """verify(vk, num_steps, z0) of a merged CycleFold proof, restated; returns the list of failed checks."""
    from vimz_amd import hip
    len_z = cf.circuit.len_z
    dg = shape_digest(cf)
    rec = parse_merged_records(merged.records(), len_z)
    failed, acc = replay_merged(orc, rec, dg, len_z)
    if acc["n"] != num_steps: failed.append("step count")
    if acc["zs"] != [int(x) for x in z0]: failed.append("z0")
    P, Q = acc["P"], acc["Q"]
    tabs = cf.r1cs(0)
    Z, E = merged.export(0, hip.IX_RUNNING_Z), merged.export(0, hip.IX_RUNNING_E)
    nw = len(Z)
    if from_limbs(Z[0:1])[0] != P[2] or from_limbs(Z[-2:]) != [P[3], P[4]]: failed.append("main: instance scalars")
    if orc.r1cs_check_relaxed(0, tabs, nw, Z, u=P[2], E=E) != -1: failed.append("main: relaxed relation")
    tabs2 = cf.r1cs(1)
    Z2, E2 = merged.export(1, hip.IX_RUNNING_Z), merged.export(1, hip.IX_RUNNING_E)
    n2 = len(Z2)
    if from_limbs(Z2[0:1])[0] != Q[2] or from_limbs(Z2[-CF_IO:]) != Q[3]: failed.append("cyclefold: instance scalars")
    if orc.r1cs_check_relaxed(1, tabs2, n2, Z2, u=Q[2], E=E2) != -1: failed.append("cyclefold: relaxed relation")
    if check_commitments:
        bases = ck1.download(0, max(nw - 3, len(E)))
        if orc.msm(0, bases[:nw - 3], Z[1:nw - 2]) != tuple(P[0]): failed.append("main: comm_W")
        if orc.msm(0, bases[:len(E)], E) != tuple(P[1]): failed.append("main: comm_E")
        bases2 = ck2.download(0, max(n2 - 1 - CF_IO, len(E2)))
        if orc.msm(1, bases2[:n2 - 1 - CF_IO], Z2[1:n2 - CF_IO]) != tuple(Q[0]): failed.append("cyclefold: comm_W")
        if orc.msm(1, bases2[:len(E2)], E2) != tuple(Q[1]): failed.append("cyclefold: comm_E")
    return failed, acc


# ---- the relation F' enforces in one step, restated natively (the counterpart of oracle/nova.hpp's nova_step for this scheme): from what the
# last step's circuit was given (vimz_cf_export VIMZ_IX_LAST_STEP) to what it must have returned ----------------------------------------------
def parse_last_step(words, len_z):
    w = [int(x) for x in np.asarray(words).reshape(-1)]
    pos = 0

    def el():
        nonlocal pos
        v = sum(w[pos + k] << (64 * k) for k in range(4))
        pos += 4
        return v
    r = {"i": el(), "z_i": [el() for _ in range(len_z)], "z_next": [el() for _ in range(len_z)], "U": [el() for _ in range(7)], "u": [el() for _ in range(4)],
         "T": (el(), el()), "Wn": (el(), el()), "En": (el(), el()), "cfU": [el() for _ in range(5 + CF_IO)]}
    r["cf1W"], r["cf1T"], r["cf2W"], r["cf2T"] = (el(), el()), (el(), el()), (el(), el()), (el(), el())
    r["U_new"], r["cfU_new"], r["x0"], r["x1"] = [el() for _ in range(7)], [el() for _ in range(5 + CF_IO)], el(), el()
    r["r"], r["r1"], r["r2"] = el(), el(), el()
    assert pos == len(w)
    return r


def step_relation(orc, dg, z0, s):
    """What F' must return for the inputs `s` (aug/cyclefold.hpp, header comment): (failed checks, U', cfU', x0, x1, challenges)."""
    pr, pq = orc.modulus[0], orc.modulus[1]
    failed = []
    base = s["i"] == 0
    U, u, cfU = s["U"], s["u"], s["cfU"]
    if base and s["z_i"] != list(z0): failed.append("base case does not start from z0")
    h_U, h_cf = hash_main(orc, dg, s["i"], z0, s["z_i"], U), hash_cf(orc, dg, cfU)
    if not base and (h_U != u[2] or h_cf != u[3]): failed.append("incoming hashes")
    hr = orc.nova_hash(0, [h_U] + limbs64(u[0]) + limbs64(u[1]) + [u[2], u[3]] + limbs64(s["T"][0]) + limbs64(s["T"][1]))
    r = chal(hr)
    h1 = orc.nova_hash(0, [h_cf, hr, s["cf1W"][0], s["cf1W"][1]] + limbs64(s["Wn"][0]) + limbs64(s["Wn"][1]) + [s["cf1T"][0], s["cf1T"][1]])
    r1 = chal(h1)
    h2 = orc.nova_hash(0, [h1, s["cf2W"][0], s["cf2W"][1]] + limbs64(s["En"][0]) + limbs64(s["En"][1]) + [s["cf2T"][0], s["cf2T"][1]])
    r2 = chal(h2)
    if base:
        Un, cn = [0] * 7, [0] * (5 + CF_IO)
        if s["Wn"] != (0, 0) or s["En"] != (0, 0): failed.append("base case hints")
    else:
        axpy = lambda cid, a, k, b: orc.curve_add(cid, a, orc.curve_mul(cid, b, k))
        # the hinted commitments are what the two CycleFold instances speak about: W' = W + r·W_in, E' = E + r·cmT on BN254 G1
        if axpy(0, (U[0], U[1]), r, (u[0], u[1])) != s["Wn"]: failed.append("hint W'")
        if axpy(0, (U[2], U[3]), r, s["T"]) != s["En"]: failed.append("hint E'")
        Un = [s["Wn"][0], s["Wn"][1], s["En"][0], s["En"][1], (U[4] + r) % pr, (U[5] + r * u[2]) % pr, (U[6] + r * u[3]) % pr]
        cf1x = [r, U[0], U[1], u[0], u[1], s["Wn"][0], s["Wn"][1]]
        cf2x = [r, U[2], U[3], s["T"][0], s["T"][1], s["En"][0], s["En"][1]]
        W = axpy(1, axpy(1, (cfU[0], cfU[1]), r1, s["cf1W"]), r2, s["cf2W"])
        E = axpy(1, axpy(1, (cfU[2], cfU[3]), r1, s["cf1T"]), r2, s["cf2T"])
        cn = [W[0], W[1], E[0], E[1], (cfU[4] + r1 + r2) % pr] + [(x + r1 * a + r2 * b) % pq for x, a, b in zip(cfU[5:], cf1x, cf2x)]
    x0 = hash_main(orc, dg, s["i"] + 1, z0, s["z_next"], Un)
    x1 = hash_cf(orc, dg, cn)
    return failed, Un, cn, x0, x1, (r, r1, r2)
