"""A restatement of the reference's on-chain Nova + CycleFold decider verifier — `NovaDecider.verifyOpaqueNovaProofWithInputs` →
`verifyNovaProof` of contracts/*Verifier.sol (ContrastVerifier.sol:685-810) — in plain Python integers on top of tests/_pairing.py.
Test infrastructure only: nothing of the product imports this.  Its purpose is to PIN the checker on reference vectors: with the constants of
the committed contracts (tests/golden/verifier_keys.json, minted by tests/golden/make_verifier_keys.py) it must accept the six committed
marketplace/proofs/*.proof (tests/test_novadecider.py), which pins the pairing, the KZG check, the Groth16 check, the public-input layout
and the order of the 25 calldata words on vectors this repository did not make.

What the contract does, line by line (all line numbers: contracts/ContrastVerifier.sol; the other eight differ only in len_z and constants):
  :785-810  verifyOpaqueNovaProofWithInputs(steps, z0, zi, proof[25]): regroup the words —
              U_i.cmW = proof[0:2], U_i.cmE = proof[2:4], u_i.cmW = proof[4:6], cmT = proof[6:8], r = proof[8],
              pA = proof[9:11], pB = [[proof[11], proof[12]], [proof[13], proof[14]]], pC = proof[15:17],
              challenge_W, challenge_E, eval_W, eval_E = proof[17:21], kzg proof_W = proof[21:23], proof_E = proof[23:25].
  :685-783  verifyNovaProof: require steps >= 2 (:697);
              public_inputs[0] = pp_hash (:703), [1] = steps (:704), [2 : 2 + 2 len_z] = z0 ‖ zi (:706-708);
              cmW = U_i.cmW + r · u_i.cmW (:712-713); its x then y as 5 limbs of 55 bits, little end first (:716-722, LimbsDecomposition :632-640);
              KZG check(cmW, proof_W, challenge_W, eval_W) (:725-730);
              cmE = U_i.cmE + r · cmT (:735-736); limbs (:739-745); KZG check(cmE, proof_E, challenge_E, eval_E) (:748-753);
              then challenge_W, challenge_E, eval_W, eval_E (:758-761) and the limbs of cmT's x and y (:763-772);
              Groth16 verifyProof(pA, pB, pC, public_inputs) (:774-775).
  :167-189  KZG10 `check(c, pi, x, y)`: e(pi, VK) · e(x · (−pi) − c + y · G_1, G_2) == 1.
  :101-139  `pairing` sends each G2 operand as (a[0][1], a[0][0], a[1][1], a[1][0]) — "imaginary part first" — so an array [[p, q], [s, t]] is the
            point x = p + q·u, y = s + t·u.
  :386-620  Groth16 `verifyProof`: every public signal < r (checkField), vk_x = IC_0 + Σ signal_k · IC_{k+1} (:436-478),
            e(−A, B) · e(alpha, beta) · e(vk_x, gamma) · e(C, delta) == 1 (:480-521); the G2 words go to the precompile in the order
            (x1, x2, y1, y2) = (imaginary, real, imaginary, real): x = x2 + x1·u.  pB is passed through as given, so proof[11] is the imaginary
            part of B.x (vimz_amd.calldata.WORD_NAMES says the same).
The EVM precompiles 6/7/8 fail (and the call reverts) on points off the curve or coordinates >= q; `verify` returns (False, reason) there."""
import json
import os

from tests import _pairing as bp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LIMB_BITS, LIMBS = 55, 5


def verifier_keys():
    with open(os.path.join(GOLDEN, "verifier_keys.json")) as fp:
        raw = json.load(fp)
    out = {}
    for name, k in raw.items():
        assert k["limb_bits"] == LIMB_BITS and k["limbs"] == LIMBS and int(k["field_r"]) == bp.R and int(k["field_q"]) == bp.Q
        g = k["groth16"]
        g2 = lambda d: ((int(d["x2"]), int(d["x1"])), (int(d["y2"]), int(d["y1"])))          # (real, imaginary) as tests/_pairing.py holds G2
        arr = lambda a: ((int(a[0][0]), int(a[0][1])), (int(a[1][0]), int(a[1][1])))       # [[p, q], [s, t]]: x = p + q u (see `pairing` above)
        out[name] = {"len_z": k["len_z"], "pp_hash": int(k["pp_hash"]),
                     "groth16": {"alpha": (int(g["alpha"][0]), int(g["alpha"][1])), "beta": g2(g["beta"]), "gamma": g2(g["gamma"]),
                                 "delta": g2(g["delta"]), "ic": [(int(x), int(y)) for x, y in g["ic"]]},
                     "kzg": {"G_1": (int(k["kzg"]["G_1"][0]), int(k["kzg"]["G_1"][1])), "G_2": arr(k["kzg"]["G_2"]), "VK": arr(k["kzg"]["VK"])}}
    return out


def limbs(x):
    """LimbsDecomposition.decompose (:632-640): five 55-bit limbs, little end first (bits above 275 are dropped, as there)."""
    return [(x >> (LIMB_BITS * i)) & ((1 << LIMB_BITS) - 1) for i in range(LIMBS)]


def _g1(p):
    """A word pair as the precompiles read it: (0, 0) is infinity; anything else must be a curve point with coordinates < q."""
    x, y = p
    if x == 0 and y == 0:
        return None
    if x >= bp.Q or y >= bp.Q or not bp.g1_on_curve((x, y)):
        raise ValueError("G1 operand is not on the curve")
    return (x, y)


def _g2(p):
    (x0, x1), (y0, y1) = p
    if x0 == x1 == y0 == y1 == 0:
        return None
    if max(x0, x1, y0, y1) >= bp.Q or not bp.g2_on_curve(p):
        raise ValueError("G2 operand is not on the twist")
    if bp.g2_mul(p, bp.R) is not None:
        raise ValueError("G2 operand is not in the order-r subgroup")      # EIP-197 requires the subgroup check
    return p


def kzg_check(kzg, c, pi, x, y):
    """KZG10Verifier.check (:167-189)."""
    c, pi = _g1(c), _g1(pi)
    rhs = bp.g1_add(bp.g1_mul(bp.g1_neg(pi), x), bp.g1_add(bp.g1_neg(c), bp.g1_mul(_g1(kzg["G_1"]), y)))
    return bp.pairing_product_is_one([(pi, _g2(kzg["VK"])), (rhs, _g2(kzg["G_2"]))])


def groth16_check(g, pA, pB, pC, public_inputs):
    """Groth16Verifier.verifyProof (:386-620); pB = [[x_imag, x_real], [y_imag, y_real]] as the calldata carries it."""
    if len(public_inputs) + 1 != len(g["ic"]) or any(not 0 <= s < bp.R for s in public_inputs):
        return False
    vk_x = _g1(g["ic"][0])
    for s, ic in zip(public_inputs, g["ic"][1:]):
        vk_x = bp.g1_add(vk_x, bp.g1_mul(_g1(ic), s))
    B = _g2(((pB[0][1], pB[0][0]), (pB[1][1], pB[1][0])))
    return bp.pairing_product_is_one([(bp.g1_neg(_g1(pA)), B), (_g1(g["alpha"]), _g2(g["beta"])), (vk_x, _g2(g["gamma"])), (_g1(pC), _g2(g["delta"]))])


def public_inputs(key, steps, z0, zi, proof):
    """The Groth16 statement `verifyNovaProof` assembles (:700-772) and the two folded commitments it opens: (list of 36 + 2 len_z ints, cmW, cmE)."""
    p = [int(w) for w in proof]
    r = p[8]
    cmW = bp.g1_add(_g1((p[0], p[1])), bp.g1_mul(_g1((p[4], p[5])), r))
    cmE = bp.g1_add(_g1((p[2], p[3])), bp.g1_mul(_g1((p[6], p[7])), r))
    xy = lambda pt: (0, 0) if pt is None else pt
    pub = [key["pp_hash"], int(steps)] + [int(v) for v in z0] + [int(v) for v in zi]
    pub += limbs(xy(cmW)[0]) + limbs(xy(cmW)[1]) + limbs(xy(cmE)[0]) + limbs(xy(cmE)[1])
    pub += [p[17], p[18], p[19], p[20]] + limbs(p[6]) + limbs(p[7])
    return pub, xy(cmW), xy(cmE)


def verify(key, steps, z0, zi, proof):
    """verifyOpaqueNovaProofWithInputs.  Returns (accepted, reason): reason names the `require` that failed, or the precompile that reverted."""
    if len(z0) != key["len_z"] or len(zi) != key["len_z"] or len(proof) != 25:
        return False, "abi: wrong number of words"
    if any(not 0 <= int(w) < 1 << 256 for w in list(z0) + list(zi) + list(proof) + [steps]):
        return False, "abi: a word does not fit 256 bits"
    if int(steps) < 2:
        return False, "Folding: the number of folded steps should be at least 2"
    p = [int(w) for w in proof]
    try:
        pub, cmW, cmE = public_inputs(key, steps, z0, zi, p)
        if not kzg_check(key["kzg"], cmW, (p[21], p[22]), p[17], p[19]):
            return False, "KZG: verifying proof for challenge W failed"
        if not kzg_check(key["kzg"], cmE, (p[23], p[24]), p[18], p[20]):
            return False, "KZG: verifying proof for challenge E failed"
        if not groth16_check(key["groth16"], (p[9], p[10]), [[p[11], p[12]], [p[13], p[14]]], (p[15], p[16]), pub):
            return False, "Groth16: verifying proof failed"
    except ValueError as e:
        return False, f"precompile reverted: {e}"
    return True, "ok"


def verify_calldata(keys, name, raw):
    """Decode a `.proof` blob (vimz_amd.calldata.decode: selector, steps, z0, zi, 25 words) and run `verify` with the contract `name`'s key."""
    from vimz_amd import calldata
    d = calldata.decode(raw)
    return verify(keys[name], d["steps"], d["z0"], d["z_i"], d["proof"])
