"""Independent verifier of a MERGED proof (vimz_ivc_merge*, vimz_amd/csrc/merge_internal.hpp): the records are replayed with Python
integers, hashlib SHA3, the oracle's Poseidon instance hash (oracle/nova.hpp) and the oracle's curve arithmetic; the two folded
instances it arrives at are then checked against the exported witnesses with the oracle's relaxed-R1CS check and MSM.  Test
infrastructure only: nothing of the product imports this."""
import hashlib

import numpy as np

from tests._oracle import from_limbs

MAGIC = 0x3147524D5A56


class _Rd:
    def __init__(self, words):
        self.w, self.pos = [int(x) for x in words], 0

    def word(self):
        self.pos += 1
        return self.w[self.pos - 1]

    def el(self):
        v = sum(self.w[self.pos + k] << (64 * k) for k in range(4))
        self.pos += 4
        return v


def parse_records(words, len_z):
    rd = _Rd(words)
    magic, S, n_ops, lz, nw1, nc1, nw2, nc2 = [rd.word() for _ in range(8)]
    assert magic == MAGIC and lz == len_z and n_ops == 2 * S - 1
    segs = []
    for _ in range(S):
        s = {"n": rd.word(), "zs": [rd.el() for _ in range(lz)], "ze": [rd.el() for _ in range(lz)]}
        s["U1"] = [rd.el() for _ in range(7)]
        s["U2"] = [rd.el() for _ in range(7)]
        s["u2"] = [rd.el() for _ in range(4)]
        s["T"] = (rd.el(), rd.el())
        segs.append(s)
    ops = []
    for _ in range(n_ops):
        kind, leaf = rd.word(), rd.word()
        if kind == 1:
            ops.append((1, leaf, (rd.el(), rd.el()), (rd.el(), rd.el())))
        else:
            assert kind == 0
            ops.append((0, leaf, None, None))
    assert rd.pos == len(rd.w)
    return {"shape": (nw1, nc1, nw2, nc2), "segs": segs, "ops": ops}


def _b(x):
    return int(x).to_bytes(32, "little")


def _chal(h, tag):
    return int.from_bytes(hashlib.sha3_256(h + tag).digest()[:16], "little")


def replay(orc, rec, digest1, digest2, len_z):
    """Returns (failed checks, accumulator): accumulator = dict(h, n, zs, ze, P, Q) with P / Q = (cW, cE, u, X0, X1)."""
    pr, pq = orc.modulus[0], orc.modulus[1]
    failed, stack, next_leaf = [], [], 0
    axpy = lambda cid, a, r, b: orc.curve_add(cid, a, orc.curve_mul(cid, b, r))
    for kind, leaf, Tp, Tq in rec["ops"]:
        if kind == 0:
            assert leaf == next_leaf
            next_leaf += 1
            s = rec["segs"][leaf]
            U1, U2, u2, T = s["U1"], s["U2"], s["u2"], s["T"]
            if orc.nova_instance_hash(0, digest1, s["n"], s["zs"], s["ze"], U2) != u2[2]:
                failed.append(f"segment {leaf}: hash of the primary chain")
            if orc.nova_instance_hash(1, digest2, s["n"], [0], [0], U1) != u2[3]:
                failed.append(f"segment {leaf}: hash of the secondary chain")
            m = b"vimz-merge-leaf-v1" + _b(digest1) + _b(digest2) + len_z.to_bytes(8, "little") + s["n"].to_bytes(8, "little")
            m += b"".join(_b(x) for x in s["zs"] + s["ze"] + U1 + U2 + u2 + list(T))
            h = hashlib.sha3_256(m).digest()
            r = _chal(h, b"q")
            P = ((U1[0], U1[1]), (U1[2], U1[3]), U1[4] % pr, U1[5] % pr, U1[6] % pr)
            Q = (axpy(1, (U2[0], U2[1]), r, (u2[0], u2[1])), axpy(1, (U2[2], U2[3]), r, T), (U2[4] + r) % pq, (U2[5] + r * u2[2]) % pq, (U2[6] + r * u2[3]) % pq)
            stack.append({"h": h, "n": s["n"], "zs": s["zs"], "ze": s["ze"], "P": P, "Q": Q})
        else:
            B, A = stack.pop(), stack.pop()
            if A["ze"] != B["zs"]:
                failed.append("segments not adjacent")
            h = hashlib.sha3_256(b"vimz-merge-node-v1" + A["h"] + B["h"] + _b(Tp[0]) + _b(Tp[1]) + _b(Tq[0]) + _b(Tq[1])).digest()
            rp, rq = _chal(h, b"p"), _chal(h, b"q")
            fold = lambda cid, X, Y, T, r, p: (axpy(cid, X[0], r, Y[0]), axpy(cid, X[1], r, axpy(cid, T, r, Y[1])), (X[2] + r * Y[2]) % p, (X[3] + r * Y[3]) % p, (X[4] + r * Y[4]) % p)
            stack.append({"h": h, "n": A["n"] + B["n"], "zs": A["zs"], "ze": B["ze"], "P": fold(0, A["P"], B["P"], Tp, rp, pr), "Q": fold(1, A["Q"], B["Q"], Tq, rq, pq)})
    assert len(stack) == 1 and next_leaf == len(rec["segs"])
    return failed, stack[0]


def verify_merged(orc, merged, vk, ck1, ck2, num_steps, z0, digest1, digest2, check_commitments=True):
    """RecursiveSNARK::verify for the merged object, restated: returns the list of failed checks."""
    from vimz_amd import hip
    len_z = vk.circuit.len_z
    rec = parse_records(merged.records(), len_z)
    failed, acc = replay(orc, rec, digest1, digest2, len_z)
    if acc["n"] != num_steps:
        failed.append("step count")
    if acc["zs"] != [int(x) for x in z0]:
        failed.append("z0")
    for side, inst, ck, cid, fid in ((0, acc["P"], ck1, 0, 0), (1, acc["Q"], ck2, 1, 1)):
        tabs = vk.r1cs(side)
        Z, E = merged.export(side, hip.IX_RUNNING_Z), merged.export(side, hip.IX_RUNNING_E)
        nw = len(Z)
        cW, cE, u, X0, X1 = inst
        if from_limbs(Z[0:1])[0] != u or from_limbs(Z[-2:]) != [X0, X1]:
            failed.append(f"side {side}: instance scalars")
        if orc.r1cs_check_relaxed(fid, tabs, nw, Z, u=u, E=E) != -1:
            failed.append(f"side {side}: relaxed relation")
        if check_commitments:
            bases = ck.download(0, max(nw - 3, len(E)))
            if orc.msm(cid, bases[:nw - 3], Z[1:nw - 2]) != tuple(cW):
                failed.append(f"side {side}: comm_W")
            if orc.msm(cid, bases[:len(E)], E) != tuple(cE):
                failed.append(f"side {side}: comm_E")
    return failed, acc
