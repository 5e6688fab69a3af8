"""The oracle-side BN254 pairing (tests/_pairing.py): the verifier of the decider's Groth16 proof must itself be right.  Bilinearity,
non-degeneracy, the group orders, and a Groth16 relation built by hand from a known trapdoor."""
from tests import _pairing as bp


def test_generators_and_orders():
    assert bp.g1_on_curve(bp.G1) and bp.g2_on_curve(bp.G2)
    assert bp.g1_mul(bp.G1, bp.R) is None and bp.g2_mul(bp.G2, bp.R) is None
    assert bp.g1_mul(bp.G1, bp.R - 1) == bp.g1_neg(bp.G1)
    x = bp.f12([3, 1, 4, 1, 5, 9, 2, 6, 5, 3, 5, 8])
    assert bp.f12_mul(x, bp.f12_inv(x)) == bp.F12_ONE


def test_pairing_is_bilinear_and_not_degenerate():
    e = bp.pairing(bp.G2, bp.G1)
    assert e != bp.F12_ONE and bp.f12_pow(e, bp.R) == bp.F12_ONE
    a, b = 0x1234567, 0x89abcdef01
    assert bp.pairing(bp.g2_mul(bp.G2, b), bp.g1_mul(bp.G1, a)) == bp.f12_pow(e, a * b)
    # the product form the EVM precompile answers: e(aP, Q) · e(-P, aQ) = 1
    assert bp.pairing_product_is_one([(bp.g1_mul(bp.G1, a), bp.G2), (bp.g1_neg(bp.G1), bp.g2_mul(bp.G2, a))])
    assert not bp.pairing_product_is_one([(bp.g1_mul(bp.G1, a), bp.G2), (bp.g1_neg(bp.G1), bp.g2_mul(bp.G2, a + 1))])


def test_groth16_equation_from_a_known_trapdoor():
    """A·B = alpha·beta + x·gamma·(ic/gamma) + C·delta in the exponent: a proof made with the trapdoor verifies, a changed input does not."""
    r = bp.R
    alpha, beta, gamma, delta = 11, 13, 17, 19
    ic0, ic1, x = 23, 29, 5          # (ic values are already divided by gamma in a real key: here they are the exponents of IC_i)
    a_, b_ = 1234, 5678
    c_ = (a_ * b_ - alpha * beta - (ic0 + x * ic1) * gamma) * pow(delta, -1, r) % r
    vk = {"alpha": bp.g1_mul(bp.G1, alpha), "beta": bp.g2_mul(bp.G2, beta), "gamma": bp.g2_mul(bp.G2, gamma), "delta": bp.g2_mul(bp.G2, delta),
          "ic": [bp.g1_mul(bp.G1, ic0), bp.g1_mul(bp.G1, ic1)]}
    proof = (bp.g1_mul(bp.G1, a_), bp.g2_mul(bp.G2, b_), bp.g1_mul(bp.G1, c_))
    assert bp.groth16_verify(vk, [x], proof)
    assert not bp.groth16_verify(vk, [x + 1], proof)
