"""The restated on-chain verifier (tests/_novadecider.py: contracts/*Verifier.sol's KZG check, limb decomposition and Groth16 check over
tests/_pairing.py) against the REFERENCE's own vectors: with the constants of the committed contracts (tests/golden/verifier_keys.json) it
accepts all six committed marketplace/proofs/*.proof and rejects each with one word changed.  This is what pins the oracle-side pairing, the
public-input layout (pp_hash, i, z0, zi, 4 x 5 limbs of the folded commitments, challenges, evaluations, 2 x 5 limbs of cmT) and the order of the
25 words on vectors this repository did not make (VERDICT r4 missing #1)."""
import pytest

from tests import _data, _novadecider as nd
from vimz_amd import calldata

# marketplace/proofs/generate-proofs.sh:72-78: which transformation (= which contract) each committed proof was made with
CONTRACT_OF = {"img1-blur": "blur", "img1-grayscale": "grayscale", "img1-sharpness-grayscale": "grayscale", "img1-sharpness": "sharpness",
               "img2-contrast-sharpness": "sharpness", "img2-contrast": "contrast"}


@pytest.fixture(scope="module")
def keys():
    return nd.verifier_keys()


def _statement(name):
    P = _data.kat()["proofs"][name]
    return P["steps"], [int(x) for x in P["z0"]], [int(x) for x in P["z_final"]], [int(x) for x in P["proof_words"]]


def test_verifier_keys_cover_the_nine_contracts(keys):
    assert sorted(keys) == ["blur", "brightness", "contrast", "crop", "grayscale", "hash", "redact", "resize", "sharpness"]
    # state widths of the reference's step circuits (vimz/src/transformation.rs:25-40) = widths of the contracts' entry points
    assert {k: v["len_z"] for k, v in keys.items()} == {"blur": 4, "sharpness": 4, "brightness": 3, "contrast": 3, "crop": 3, "grayscale": 2,
                                                        "redact": 2, "resize": 2, "hash": 1}
    for k, v in keys.items():
        assert len(v["groth16"]["ic"]) == 1 + 36 + 2 * v["len_z"]
        assert all(nd.bp.g1_on_curve(p) for p in v["groth16"]["ic"] + [v["groth16"]["alpha"], v["kzg"]["G_1"]])
        assert all(nd.bp.g2_on_curve(v["groth16"][g]) for g in ("beta", "gamma", "delta")) and nd.bp.g2_on_curve(v["kzg"]["G_2"]) and nd.bp.g2_on_curve(v["kzg"]["VK"])
    # every contract was generated over the same KZG SRS (StdRng::from_seed([41; 32]), vimz/src/sonobe_backend/mod.rs:54)
    assert len({str(v["kzg"]) for v in keys.values()}) == 1


@pytest.mark.parametrize("name", sorted(CONTRACT_OF))
def test_the_reference_s_committed_proofs_are_accepted(keys, name):
    steps, z0, zf, words = _statement(name)
    assert nd.verify(keys[CONTRACT_OF[name]], steps, z0, zf, words) == (True, "ok")
    # ... through the calldata bytes as well, and NOT by a contract of the same width made for another circuit
    raw = calldata.encode(steps, z0, zf, words)
    assert nd.verify_calldata(keys, CONTRACT_OF[name], raw) == (True, "ok")
    other = {"blur": "sharpness", "sharpness": "blur", "grayscale": "resize", "contrast": "crop"}[CONTRACT_OF[name]]
    assert nd.verify(keys[other], steps, z0, zf, words) == (False, "Groth16: verifying proof failed")


def test_every_word_of_a_committed_proof_matters(keys):
    """img2-contrast.proof with each of its 1 + 3 + 3 + 25 words changed in turn: rejected, and by the check the contract would fail in."""
    steps, z0, zf, words = _statement("img2-contrast")
    key = keys["contrast"]
    g16 = "Groth16: verifying proof failed"
    assert nd.verify(key, steps + 1, z0, zf, words) == (False, g16)
    assert nd.verify(key, 1, z0, zf, words)[0] is False
    for k in range(3):
        assert nd.verify(key, steps, z0[:k] + [z0[k] ^ 1] + z0[k + 1:], zf, words) == (False, g16)
        assert nd.verify(key, steps, z0, zf[:k] + [zf[k] ^ 1] + zf[k + 1:], words) == (False, g16)
    for k in range(25):
        bad = list(words)
        bad[k] ^= 1
        ok, why = nd.verify(key, steps, z0, zf, bad)
        assert not ok, calldata.WORD_NAMES[k]
        if k in (17, 19):
            assert why.startswith("KZG: verifying proof for challenge W"), (k, why)
        elif k in (18, 20):
            assert why.startswith("KZG: verifying proof for challenge E"), (k, why)
        elif k == 8:
            assert why.startswith("KZG"), (k, why)                  # r moves both folded commitments
        else:
            assert why.startswith("precompile reverted"), (k, why)  # a coordinate changed by one is off the curve
    # points that stay ON the curve but are wrong: negate them (y -> q - y)
    q = nd.bp.Q
    for iy, expect in ((1, "KZG: verifying proof for challenge W"), (3, "KZG: verifying proof for challenge E"), (5, "KZG: verifying proof for challenge W"),
                       (7, "KZG: verifying proof for challenge E"), (10, g16), (16, g16), (22, "KZG: verifying proof for challenge W"),
                       (24, "KZG: verifying proof for challenge E")):
        bad = list(words)
        bad[iy] = q - bad[iy]
        ok, why = nd.verify(key, steps, z0, zf, bad)
        assert not ok and why.startswith(expect), (calldata.WORD_NAMES[iy], why)
    bad = list(words)                                                # B negated: both y words
    bad[13], bad[14] = q - bad[13], q - bad[14]
    assert nd.verify(key, steps, z0, zf, bad) == (False, g16)
    bad = list(words)                                                # B's real and imaginary parts swapped: not on the twist
    bad[11], bad[12], bad[13], bad[14] = bad[12], bad[11], bad[14], bad[13]
    assert nd.verify(key, steps, z0, zf, bad)[1].startswith("precompile reverted")


@pytest.mark.parametrize("name", sorted(set(CONTRACT_OF) - {"img2-contrast"}))
def test_the_other_committed_proofs_reject_a_changed_statement_and_a_changed_opening(keys, name):
    steps, z0, zf, words = _statement(name)
    key = keys[CONTRACT_OF[name]]
    assert nd.verify(key, steps, z0, zf[:-1] + [zf[-1] ^ 1], words) == (False, "Groth16: verifying proof failed")
    bad = list(words)
    bad[19] = (bad[19] + 1) % nd.bp.R
    assert nd.verify(key, steps, z0, zf, bad)[1].startswith("KZG: verifying proof for challenge W")
    bad = list(words)
    bad[24] = nd.bp.Q - bad[24]
    assert nd.verify(key, steps, z0, zf, bad)[1].startswith("KZG: verifying proof for challenge E")


def test_limb_decomposition_is_the_contract_s():
    x = (0x1234567 << 220) | (0x5A5A5A5A5A5A5 << 110) | 0x7FFFFFFFFFFFFF
    l = nd.limbs(x)
    assert len(l) == 5 and all(v < 1 << 55 for v in l) and sum(v << (55 * i) for i, v in enumerate(l)) == x
    assert nd.limbs((1 << 256) - 1) == [(1 << 55) - 1] * 4 + [(1 << 36) - 1]      # 256 = 4 * 55 + 36
