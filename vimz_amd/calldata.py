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
on marketplace/proofs).  The 25 proof words are Sonobe's Nova+CycleFold instance commitments over a KZG SRS drawn from
`StdRng::from_seed([41; 32])` (sonobe_backend/mod.rs:54) and a Groth16 decider proof: they cannot be produced without that SRS and
the decider circuit's proving key, neither of which exists in this repository or image.  So `encode` takes the 25 words from the
caller; this library's own proofs (vimz_ivc_*, vimz_ivc_merge*, vimz_cf_*) are NOT accepted by those contracts.  `decider_words` fills the 17
words that do not come from the Groth16 prover for a proof of this library's Nova + CycleFold scheme: what of
`Decider::prove` the GPU pipeline covers — the final fold and the KZG openings —, in our protocol's values.
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


GROTH16_WORDS = list(range(9, 17))      # the eight words of the decider's Groth16 proof (decider_words(..., decider=...) fills them)


def decider_public_hash_inputs(words):
    """The values h_inst binds (vimz_amd/csrc/aug/decider.hpp), in its order, from the 25 words: rho, then the six commitments' 64-bit limbs
    (U_i.cmW, U_i.cmE, u_i.cmW, cmT, and U_{i+1}.cmW / cmE which the verifier computes itself), then the four KZG scalars."""
    return {"rho": words[8], "points": [(words[0], words[1]), (words[2], words[3]), (words[4], words[5]), (words[6], words[7])], "kzg": [words[17], words[18], words[19], words[20]]}


def decider_words(cf_prover, decider=None):
    """The 17 of the 25 proof words that do not come out of the Groth16 prover — and, given a vimz_amd.hip.Decider set up over the same
    prover, the eight that do (then also returns its public inputs as a third value) —, for a Nova + CycleFold proof made by `cf_prover`
    (vimz_amd.hip.CycleFoldIVC, its commitment key being the KZG SRS's powers) — in THIS library's protocol, so a statement of
    what the GPU pipeline covers of `Decider::prove` (vimz/src/sonobe_backend/decider.rs:13-21), not bytes a contract generated for Sonobe accepts:
        U_i.cmW, U_i.cmE, u_i.cmW                       the running and the last instance's commitments
        cmT, r                                          the decider's final fold U_{i+1} = NIFS(U_i, u_i): cross-term commitment and challenge
        kzg.challenge / eval / proof (W, E)             openings of U_{i+1}'s two commitments at challenges derived from them
    Returns a list of 25 entries (ints; None in the Groth16 slots) and the folded commitments (U_{i+1}.cmW, U_{i+1}.cmE)."""
    import hashlib
    import numpy as np
    from . import _lib
    from .hip import CycleFoldMerged, IX_INSTANCE
    r_mod = _lib.MODULUS[0]
    m = CycleFoldMerged(cf_prover)              # one segment: acc = U_i (+) u_i
    try:
        w = [int(x) for x in m.records()]
        lz = w[2]
        pos = 8 + w[7] + 1 + 8 * lz               # header, run starts, n, z_start, z_end
        el = lambda k: sum(w[pos + 4 * k + q] << (64 * q) for q in range(4))
        U, u = [el(k) for k in range(7)], [el(7 + k) for k in range(4)]
        T2 = (el(7 + 4 + 12 + 2), el(7 + 4 + 12 + 3))
        folded = [sum(int(a[q]) << (64 * q) for q in range(4)) for a in np.asarray(m.export(0, IX_INSTANCE))]
        # r as the merge derives it: recomputed by the verifier from the records (cyclefold_merge.hip: cfm_challenges); taken from u' - U.u here
        r = (folded[4] - U[4]) % r_mod
        words = [None] * PROOF_WORDS
        words[0:4] = U[0:4]
        words[4:6] = u[0:2]
        words[6:8] = list(T2)
        words[8] = r
        for k, which in enumerate((0, 1)):
            comm = (folded[2 * which], folded[2 * which + 1])
            ch = int.from_bytes(hashlib.sha3_256(b"vimz-kzg-challenge" + comm[0].to_bytes(32, "little") + comm[1].to_bytes(32, "little")).digest(), "little") % r_mod
            ev, proof = m.kzg_open(which, ch)
            words[17 + k], words[19 + k] = ch, ev
            words[21 + 2 * k], words[22 + 2 * k] = proof
        if decider is not None:      # Decider::prove: the Groth16 proof of the final fold (A; B with the imaginary parts first, as the EVM's precompile takes G2; C)
            pub, (A, B, Cp), _ = decider.prove(m, (words[17], words[18], words[19], words[20]))
            words[9:11] = list(A)
            words[11:15] = [B[0][1], B[0][0], B[1][1], B[1][0]]
            words[15:17] = list(Cp)
            return words, ((folded[0], folded[1]), (folded[2], folded[3])), pub
        return words, ((folded[0], folded[1]), (folded[2], folded[3]))
    finally:
        m.close()


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
