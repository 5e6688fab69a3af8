"""The PRODUCT's decider verifier (vimz_decider_verify_key: vimz_amd/csrc/groth16.hip over pairing.hpp — host code, no GPU) on the reference's own
vectors: with the constants of contracts/*Verifier.sol as its key it accepts the six committed marketplace/proofs/*.proof and rejects them with one
word changed — the same verdicts as the restated contract (tests/_novadecider.py).  This pins the product's pairing, KZG check, Groth16 check, limb
decomposition and word order on bytes this repository did not make (`Decider::verify`, vimz/src/sonobe_backend/decider.rs:31-50)."""
import pytest

from tests import _novadecider as nd
from tests.test_novadecider import CONTRACT_OF, _statement
from vimz_amd import _lib
from vimz_amd.hip import decider_verify_key, parse_verifying_key, verifying_key_words


@pytest.fixture(scope="module")
def keys():
    return nd.verifier_keys()


@pytest.mark.parametrize("name", sorted(CONTRACT_OF))
def test_the_product_verifier_accepts_the_reference_s_committed_proofs(keys, name):
    steps, z0, zf, words = _statement(name)
    key = keys[CONTRACT_OF[name]]
    kw = verifying_key_words(key)
    assert parse_verifying_key(kw) == {k: key[k] for k in ("len_z", "pp_hash", "groth16", "kzg")}      # (the word layout round-trips)
    assert decider_verify_key(kw, steps, z0, zf, words) == 0
    # one changed word each: the bit the product reports is the `require` the restated contract fails in
    q, r = nd.bp.Q, nd.bp.R
    for what, change, bit in (("z_i", None, 8), ("eval_W", (19, lambda v: (v + 1) % r), 2), ("proof_E", (24, lambda v: q - v), 4), ("C", (16, lambda v: q - v), 8)):
        w, zi = list(words), list(zf)
        if change is None:
            zi[-1] ^= 1
        else:
            w[change[0]] = change[1](w[change[0]])
        got = decider_verify_key(kw, steps, z0, zi, w)
        ok, why = nd.verify(key, steps, z0, zi, w)
        assert got & bit and not ok, (what, got, why)
    w = list(words); w[4] ^= 1                                   # u_i.cmW off the curve
    assert decider_verify_key(kw, steps, z0, zf, w) & 16
    assert decider_verify_key(kw, 1, z0, zf, words) & 1
    # a key for another circuit of the same width
    other = {"blur": "sharpness", "sharpness": "blur", "grayscale": "resize", "contrast": "crop"}[CONTRACT_OF[name]]
    assert decider_verify_key(verifying_key_words(keys[other]), steps, z0, zf, words) == 8


def test_malformed_keys_are_refused(keys):
    steps, z0, zf, words = _statement("img2-contrast")
    kw = verifying_key_words(keys["contrast"])
    for bad in (kw[:-1], kw[:40]):
        with pytest.raises(_lib.VimzError):
            decider_verify_key(bad, steps, z0, zf, words)
    k2 = kw.copy(); k2[5 + 1] ^= 1                               # alpha.x: not on the curve any more
    with pytest.raises(_lib.VimzError):
        decider_verify_key(k2, steps, z0, zf, words)
    with pytest.raises(_lib.VimzError):
        decider_verify_key(kw, steps, z0[:2], zf[:2], words)     # the key is for a state of three elements
