"""Calldata layout of the on-chain verifiers (SURVEY.md §8f row N1 — the part of the Sonobe path that can be checked in this image).

`vimz -b sonobe` ends in `prepare_contract_calldata` (vimz/src/sonobe_backend/solidity.rs:13-27): sonobe_solidity's
`prepare_calldata_for_nova_cyclefold_verifier(OpaqueWithInputs, i, z_0, z_i, U_i, u_i, proof)`, the bytes
`contracts/*Verifier.sol::verifyOpaqueNovaProofWithInputs(uint256 steps, uint256[N] initial_state, uint256[N] final_state,
uint256[25] proof)` takes (ContrastVerifier.sol:785-810) and marketplace/proofs/*.proof hold
(marketplace/vimz_marketplace_sdk/artifacts.py:19-42 parses them):

    4-byte selector ‖ steps ‖ z_0[N] ‖ z_i[N] ‖ 25 words, every word a 32-byte BIG-endian integer;
    the 25 words = U_i.cmW (x, y), U_i.cmE (x, y), u_i.cmW (x, y), cmT (x, y), r, Groth16 A (x, y), B (x1, x0, y1, y0), C (x, y),
                   KZG challenges (W, E), KZG evaluations (W, E), KZG proofs (W: x, y; E: x, y).

What this module is and is not.  The statement part — selector, steps, z_0, z_i — is the SAME for every backend: the GPU prover's
(steps, z_0, z_n) for an image are bit-identical to the committed proofs' (tests/test_gpu_fold.py, tests/test_gpu_merge.py pin that
on marketplace/proofs).  The 25 proof words of the COMMITTED proofs are Sonobe's Nova+CycleFold instance commitments over a KZG SRS drawn
from `StdRng::from_seed([41; 32])` (sonobe_backend/mod.rs:54) and a Groth16 proof under Sonobe's decider key: they cannot be reproduced
without those keys.  This library's decider (vimz_decider_*, vimz_amd.hip.Decider) fills the same 25 words for ITS circuit and ITS keys,
with the public-input layout of the reference's contracts: a contract generated from the reference's template with this library's
verifying key as constants accepts them — which tests/_novadecider.py (the restated contract, pinned on the six committed proofs) checks.
"""

MASK64 = (1 << 64) - 1
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44]
_PIL = [10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1]


def _rotl(x, n):
    return ((x << n) | (x >> (64 - n))) & MASK64


def _keccak_f(st):
    for rc in _RC:
        bc = [st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20] for i in range(5)]
        for i in range(5):
            t = bc[(i + 4) % 5] ^ _rotl(bc[(i + 1) % 5], 1)
            for j in range(0, 25, 5):
                st[j + i] ^= t
        t = st[1]
        for i in range(24):
            j = _PIL[i]
            st[j], t = _rotl(t, _ROT[i]), st[j]
        for j in range(0, 25, 5):
            row = st[j:j + 5]
            for i in range(5):
                st[j + i] = row[i] ^ ((~row[(i + 1) % 5]) & MASK64 & row[(i + 2) % 5])
        st[0] ^= rc


def keccak256(data):
    """Ethereum's Keccak-256 (padding 0x01, not SHA3's 0x06) — for the 4-byte function selector."""
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    st = [0] * 25
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            st[i] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
        _keccak_f(st)
    return b"".join(x.to_bytes(8, "little") for x in st[:4])


PROOF_WORDS = 25
WORD_NAMES = ["U_i.cmW.x", "U_i.cmW.y", "U_i.cmE.x", "U_i.cmE.y", "u_i.cmW.x", "u_i.cmW.y", "cmT.x", "cmT.y", "r",
              "groth16.A.x", "groth16.A.y", "groth16.B.x1", "groth16.B.x0", "groth16.B.y1", "groth16.B.y0", "groth16.C.x", "groth16.C.y",
              "kzg.challenge_W", "kzg.challenge_E", "kzg.eval_W", "kzg.eval_E", "kzg.proof_W.x", "kzg.proof_W.y", "kzg.proof_E.x", "kzg.proof_E.y"]
G1_POINTS = [(0, 1), (2, 3), (4, 5), (6, 7), (9, 10), (15, 16), (21, 22), (23, 24)]      # word indices of the BN254 G1 points


GROTH16_WORDS = list(range(9, 17))      # the eight words of the decider's Groth16 proof


def decider_calldata(decider, ivc=None):
    """`Decider::prove` + `prepare_contract_calldata` (vimz/src/sonobe_backend/mod.rs:76-78, solidity.rs:13-27) for the IVC proof `ivc` (default: the
    prover `decider` — a vimz_amd.hip.Decider — was set up over) holds: (calldata bytes, {steps, z0, z_i, words, public_inputs, seconds})."""
    ivc = decider.prover if ivc is None else ivc
    words, pub, sec = decider.prove(ivc)
    lz = ivc.len_z if hasattr(ivc, "len_z") else (len(pub) - 36) // 2
    steps, z0, z_i = pub[1], pub[2:2 + lz], pub[2 + lz:2 + 2 * lz]
    return encode(steps, z0, z_i, words), {"steps": steps, "z0": z0, "z_i": z_i, "words": words, "public_inputs": pub, "seconds": sec}


def selector(len_z):
    """First four bytes of keccak256 of the verifier entry point's signature for a state of len_z elements."""
    sig = f"verifyOpaqueNovaProofWithInputs(uint256,uint256[{len_z}],uint256[{len_z}],uint256[{PROOF_WORDS}])"
    return keccak256(sig.encode())[:4]


def encode(steps, z0, z_i, proof_words):
    """The calldata bytes for (steps, z_0, z_i, 25 proof words)."""
    z0, z_i, proof_words = [int(x) for x in z0], [int(x) for x in z_i], [int(x) for x in proof_words]
    if len(z0) != len(z_i) or len(proof_words) != PROOF_WORDS:
        raise ValueError("calldata: z_0 and z_i must have the same length and the proof 25 words")
    words = [int(steps)] + z0 + z_i + proof_words
    if any(w < 0 or w >> 256 for w in words):
        raise ValueError("calldata: a word does not fit 256 bits")
    return selector(len(z0)) + b"".join(w.to_bytes(32, "big") for w in words)


def decode(raw):
    """{selector, len_z, steps, z0, z_i, proof: [25 ints], named: {word name: int}} of a calldata blob (what artifacts.py's
    ProofData reads); raises ValueError when the length or the selector does not fit the layout."""
    raw = bytes(raw)
    body = raw[4:]
    if len(raw) < 4 + 32 * (1 + PROOF_WORDS) or len(body) % 32 or ((len(body) // 32 - 1 - PROOF_WORDS) % 2):
        raise ValueError("calldata: not a whole number of words around a 25-word proof")
    n = (len(body) // 32 - 1 - PROOF_WORDS) // 2
    if raw[:4] != selector(n):
        raise ValueError(f"calldata: selector {raw[:4].hex()} is not verifyOpaqueNovaProofWithInputs for a state of {n} elements")
    w = [int.from_bytes(body[s:s + 32], "big") for s in range(0, len(body), 32)]
    proof = w[1 + 2 * n:]
    return {"selector": raw[:4].hex(), "len_z": n, "steps": w[0], "z0": w[1:1 + n], "z_i": w[1 + n:1 + 2 * n], "proof": proof,
            "named": dict(zip(WORD_NAMES, proof))}
