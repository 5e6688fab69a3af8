"""The calldata layout of the reference's on-chain verifiers (vimz/src/sonobe_backend/solidity.rs:13-27,
marketplace/vimz_marketplace_sdk/artifacts.py:19-42, contracts/ContrastVerifier.sol:785-810) against the six committed
marketplace/proofs/*.proof files (decoded into tests/golden/kat.json by tests/golden/make_fixtures.py): selectors, round trip,
statement parts, and — which pins the ORDER of the 25 words — every word pair this layout calls a G1 point is on BN254 G1."""
import pytest

from tests import _data
from vimz_amd import calldata


def test_selectors_are_the_keccak_of_the_entry_point_signature():
    assert calldata.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"      # Keccak-256 KAT
    # e2fd1766 (len(z) = 4), 165cb388 (2), 5e342634 (3)
    assert [calldata.selector(n).hex() for n in (4, 2, 3)] == ["e2fd1766", "165cb388", "5e342634"]


@pytest.mark.parametrize("name", ["img1-blur", "img1-grayscale", "img1-sharpness-grayscale", "img1-sharpness", "img2-contrast-sharpness", "img2-contrast"])
def test_committed_proofs_round_trip_and_their_points_are_on_the_curve(oracle, name):
    P = _data.kat()["proofs"][name]
    z0, zf, words = [int(x) for x in P["z0"]], [int(x) for x in P["z_final"]], [int(x) for x in P["proof_words"]]
    raw = calldata.encode(P["steps"], z0, zf, words)
    assert raw[:4].hex() == P["selector"] and len(raw) == 4 + 32 * (1 + 2 * len(z0) + 25)
    d = calldata.decode(raw)
    assert (d["steps"], d["z0"], d["z_i"], d["proof"], d["len_z"]) == (720, z0, zf, words, len(z0))
    for ix, iy in calldata.G1_POINTS:                     # U_i.cmW, U_i.cmE, u_i.cmW, cmT, Groth16 A and C, both KZG proofs
        assert oracle.on_curve(0, (words[ix], words[iy])), (name, calldata.WORD_NAMES[ix])
    # ... and the four words between Groth16's A and C are a point of BN254 G2 in the EVM's order (imaginary parts first): x = x0 + x1 u
    from tests import _pairing as bp
    B = ((d["named"]["groth16.B.x0"], d["named"]["groth16.B.x1"]), (d["named"]["groth16.B.y0"], d["named"]["groth16.B.y1"]))
    assert bp.g2_on_curve(B), name
    assert not bp.g2_on_curve(((B[0][1], B[0][0]), (B[1][1], B[1][0])))      # (the other order is not on the twist: the layout's order is pinned)
    assert 0 < d["named"]["r"] < 1 << 128                 # the folding challenge is a 128-bit value
    q = oracle.modulus[0]
    assert all(d["named"][k] < q for k in ("kzg.challenge_W", "kzg.challenge_E", "kzg.eval_W", "kzg.eval_E"))


def test_malformed_calldata_is_refused():
    P = _data.kat()["proofs"]["img2-contrast"]
    raw = calldata.encode(P["steps"], P["z0"], P["z_final"], P["proof_words"])
    for bad in (raw[:-1], raw[:4 + 32 * 20], b"\0\0\0\0" + raw[4:], raw + b"\0" * 32):
        with pytest.raises(ValueError):
            calldata.decode(bad)
    with pytest.raises(ValueError):
        calldata.encode(1, [0], [0, 0], [0] * 25)
    with pytest.raises(ValueError):
        calldata.encode(1, [0], [0], [0] * 24)
