"""ORACLE — TEST INFRASTRUCTURE ONLY.  An independent verifier of the product's compressed proof (CompressedSNARK::verify,
vimz/src/nova_snark_backend/mod.rs:63-67), written from the protocol description at the top of vimz_amd/csrc/spartan.hip and
sharing no code with it: Python integers for the fields, hashlib for the SHA3-256 transcript, the CPU oracle's curve
arithmetic and MSM (oracle/curve.hpp) for the group side, the exported R1CS tables for the sparse evaluation.  Parity of this
piece is UNPINNED like the augmented circuits: nova-snark 0.23.0 is not vendored and the reference holds no vector for its
CompressedSNARK; what this checks is that the product's prover convinces a second, separately written verifier of the
documented relation (and that tampering is caught).

Pure-Python loops: meant for the small shapes of the tests (2^13 .. 2^15 rows); a 2^19-row proof takes minutes here."""
import hashlib

from tests._oracle import CURVE_BASE, CURVE_SCALAR, ck_derive, from_limbs, to_limbs

MAGIC = 0x314e5343565a
M128 = (1 << 128) - 1


class Transcript:
    def __init__(self, label):
        self.st = bytes(32)
        self.absorb(b"init", label)

    def absorb(self, tag, data):
        self.st = hashlib.sha3_256(self.st + tag[:8].ljust(8, b"\0") + len(data).to_bytes(8, "little") + data).digest()

    def fe(self, tag, x):
        self.absorb(tag, int(x).to_bytes(32, "little"))

    def challenge(self):
        self.st = hashlib.sha3_256(self.st + b"c").digest()
        return int.from_bytes(self.st[:16], "little")


class Reader:
    def __init__(self, blob):
        self.b, self.pos = bytes(blob), 0

    def word(self):
        v = int.from_bytes(self.b[self.pos:self.pos + 8], "little"); self.pos += 8
        return v

    def fe(self, p=None):
        v = int.from_bytes(self.b[self.pos:self.pos + 32], "little"); self.pos += 32
        if p is not None and v >= p:
            raise ValueError("element not below its modulus")
        return v

    def point(self, p):
        return (self.fe(p), self.fe(p))


def eq_table(pt, p):
    """eq(pt, ·) with pt[0] the top bit of the index."""
    tab = [1]
    for r in reversed(pt):
        tab = [t * (1 - r) % p for t in tab] + [t * r % p for t in tab]
    return tab


def interp(ys, r, p):
    """Lagrange interpolation through (0, ys[0]), (1, ys[1]), ... evaluated at r."""
    n, acc = len(ys), 0
    for i, y in enumerate(ys):
        num, den = 1, 1
        for j in range(n):
            if j != i:
                num = num * (r - j) % p
                den = den * (i - j) % p
        acc = (acc + y * num * pow(den, -1, p)) % p
    return acc


def _ipa_verify(orc, cid, tr, Ugen, bases, b, comm, claim, rd, pb, ps):
    tr.fe(b"ipaP.x", comm[0]); tr.fe(b"ipaP.y", comm[1]); tr.fe(b"ipaC", claim)
    Up = orc.curve_mul(cid, Ugen, tr.challenge())
    P = orc.curve_add(cid, comm, orc.curve_mul(cid, Up, claim))
    rounds = (len(b) - 1).bit_length()
    assert len(b) == 1 << rounds == len(bases)
    xs = []
    for _ in range(rounds):
        L, R = rd.point(pb), rd.point(pb)
        tr.fe(b"L.x", L[0]); tr.fe(b"L.y", L[1]); tr.fe(b"R.x", R[0]); tr.fe(b"R.y", R[1])
        x = tr.challenge()
        xs.append(x)
        P = orc.curve_add(cid, orc.curve_add(cid, R, orc.curve_mul(cid, P, x)), orc.curve_mul(cid, L, x * x % ps))
    a = rd.fe(ps)
    tr.fe(b"a", a)
    s = [1]
    for x in reversed(xs):                      # index bit of round 0 = top bit
        s = s + [v * x % ps for v in s]
    b_fin = sum(u * v for u, v in zip(s, b)) % ps
    G_fin = orc.msm(cid, bases, to_limbs(s))
    rhs = orc.curve_add(cid, orc.curve_mul(cid, G_fin, a), orc.curve_mul(cid, Up, a * b_fin % ps))
    return P == rhs


def _spartan_verify(orc, cid, tr, digest, inst, tabs, n_w, n_c, key, Ugen, side_tag, rd, has_E):
    """inst = (cW, cE, u, X0, X1).  tabs: exported R1CS (CSR + canonical dictionary).  key: (n, 8) limbs of the commitment key."""
    ps, pb = orc.modulus[CURVE_SCALAR[cid]], orc.modulus[CURVE_BASE[cid]]
    cW, cE, u, X0, X1 = inst
    tr.absorb(b"side", side_tag.to_bytes(8, "little"))
    tr.fe(b"digest", digest)
    tr.fe(b"cWx", cW[0]); tr.fe(b"cWy", cW[1]); tr.fe(b"cEx", cE[0]); tr.fe(b"cEy", cE[1]); tr.fe(b"u", u); tr.fe(b"X0", X0); tr.fe(b"X1", X1)
    s, t = (n_c - 1).bit_length(), (n_w - 1).bit_length()
    tau = [tr.challenge() for _ in range(s)]
    claim, rx = 0, []
    for _ in range(s):
        s0, s2, s3 = rd.fe(ps), rd.fe(ps), rd.fe(ps)
        tr.fe(b"o0", s0); tr.fe(b"o2", s2); tr.fe(b"o3", s3)
        r = tr.challenge(); rx.append(r)
        claim = interp([s0, (claim - s0) % ps, s2, s3], r, ps)
    va, vb, vc, ve = rd.fe(ps), rd.fe(ps), rd.fe(ps), rd.fe(ps)
    if not has_E and ve != 0:
        return "E claim of a strict instance"
    e = 1
    for a_, b_ in zip(tau, rx):
        e = e * (a_ * b_ + (1 - a_) * (1 - b_)) % ps
    if e * (va * vb - u * vc - ve) % ps != claim:
        return "outer sum-check"
    for v in (va, vb, vc, ve):
        tr.fe(b"claim", v)
    rho = tr.challenge()
    claim, ry = (va + rho * vb + rho * rho * vc) % ps, []
    for _ in range(t):
        s0, s2 = rd.fe(ps), rd.fe(ps)
        tr.fe(b"i0", s0); tr.fe(b"i2", s2)
        r = tr.challenge(); ry.append(r)
        claim = interp([s0, (claim - s0) % ps, s2], r, ps)
    evalW = rd.fe(ps)
    tr.fe(b"evalW", evalW)
    ex, ey = eq_table(rx, ps), eq_table(ry, ps)
    dic = from_limbs(tabs["dict_canon"])
    vm = 0
    for m, w in zip("ABC", (1, rho, rho * rho % ps)):
        rp, col, coef = tabs[f"{m}_rowptr"], tabs[f"{m}_col"], tabs[f"{m}_coef"]
        acc = 0
        for r in range(len(rp) - 1):
            lo, hi = int(rp[r]), int(rp[r + 1])
            if hi > lo:
                acc += ex[r] * sum(dic[int(coef[k])] * ey[int(col[k])] for k in range(lo, hi))
        vm = (vm + w * acc) % ps
    vz = (evalW + u * ey[0] + X0 * ey[n_w - 2] + X1 * ey[n_w - 1]) % ps
    if vm * vz % ps != claim:
        return "inner sum-check"
    N, Mr = 1 << t, 1 << s
    shifted = [key[N - 1]] + list(key[:N - 1])          # index i (wire i) -> ck[i - 1], index 0 -> ck[N - 1]
    import numpy as np
    # the W opening is over the committed positions only: the public slots (wire 0, the last two wires) and the padding above the
    # last wire are zeroed in b, so nothing a prover puts under those generators can stand in for (u, X0, X1)
    eyW = [0 if (i == 0 or i + 2 >= n_w) else v for i, v in enumerate(ey)]
    if not _ipa_verify(orc, cid, tr, Ugen, np.array(shifted), eyW, cW, evalW, rd, pb, ps):
        return "opening of W"
    if has_E and not _ipa_verify(orc, cid, tr, Ugen, np.array(key[:Mr]), ex, cE, ve, rd, pb, ps):
        return "opening of E"
    return None


def verify_compressed(orc, blob, num_steps, z0, digest1, digest2, tabs1, tabs2, n1, n2, key1, key2):
    """Returns a list of failed checks ([] = accepted).  n1 = (wires, constraints) of the primary augmented circuit, n2 of the
    secondary; key1 / key2: commitment keys as (n, 8) uint64 canonical affine limbs; digests: the verifier's own shape digests."""
    failed = []
    pr, pq = orc.modulus[0], orc.modulus[1]
    rd = Reader(blob)
    magic, steps, lz, s1, t1, s2, t2, _ = [rd.word() for _ in range(8)]
    if magic != MAGIC or (s1, t1, s2, t2) != ((n1[1] - 1).bit_length(), (n1[0] - 1).bit_length(), (n2[1] - 1).bit_length(), (n2[0] - 1).bit_length()):
        return ["header"], None
    z0p = [rd.fe(pr) for _ in range(lz)]
    zn = [rd.fe(pr) for _ in range(lz)]
    if steps != num_steps or z0p != [int(x) for x in z0]:
        failed.append("statement")
    inst_lo = rd.pos
    U1 = [rd.fe(pq) for _ in range(5)] + [rd.fe(), rd.fe()]       # coordinates / u in Fq, X0 / X1 256-bit integers
    U2 = [rd.fe(pr) for _ in range(5)] + [rd.fe(), rd.fe()]
    u2 = [rd.fe(pr) for _ in range(4)]
    inst_bytes = rd.b[inst_lo:rd.pos]
    if orc.nova_instance_hash(0, digest1, steps, list(z0), zn, U2) != u2[2]:
        failed.append("hash of the primary chain")
    if orc.nova_instance_hash(1, digest2, steps, [0], [0], U1) != u2[3]:
        failed.append("hash of the secondary chain")
    tr = Transcript(b"vimz-compressed-snark-v1")
    tr.absorb(b"steps", steps.to_bytes(8, "little"))
    for z in z0p:
        tr.fe(b"z0", z)
    for z in zn:
        tr.fe(b"zn", z)
    tr.absorb(b"inst", inst_bytes)
    Ug1, Ug2 = ck_derive(0, b"vimz-ipa-u", 0), ck_derive(1, b"vimz-ipa-u", 0)
    i1 = ((U1[0], U1[1]), (U1[2], U1[3]), U1[4], U1[5] % pr, U1[6] % pr)
    i2 = ((U2[0], U2[1]), (U2[2], U2[3]), U2[4], U2[5] % pq, U2[6] % pq)
    i3 = ((u2[0], u2[1]), (0, 0), 1, u2[2], u2[3])
    for name, cid, dg, inst, tabs, n, key, Ug, tag, has_E in (("primary", 0, digest1, i1, tabs1, n1, key1, Ug1, 1, True),
                                                               ("secondary", 1, digest2, i2, tabs2, n2, key2, Ug2, 2, True),
                                                               ("fresh secondary", 1, digest2, i3, tabs2, n2, key2, Ug2, 3, False)):
        why = _spartan_verify(orc, cid, tr, dg, inst, tabs, n[0], n[1], key, Ug, tag, rd, has_E)
        if why:
            failed.append(f"{name}: {why}")
            return failed, zn
    if rd.pos != len(rd.b):
        failed.append("trailing bytes")
    return failed, zn
