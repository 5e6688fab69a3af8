"""The decider of the Nova + CycleFold path (vimz_decider_*, vimz_amd/csrc/groth16.hip, aug/decider.hpp): `Decider::preprocess` / `prove` of the
reference's Sonobe backend (vimz/src/sonobe_backend/mod.rs:72-78, decider.rs:13-21) — Groth16 over BN254 for OUR statement of the final fold, with a
deterministic test setup.  The verifier here is the oracle-side pairing (tests/_pairing.py: what the EVM's precompile behind contracts/*Verifier.sol
computes): it accepts the proof for exactly the public inputs the 25 calldata words imply, and nothing else."""
import numpy as np
import pytest

from tests import _pairing as bp
from tests._oracle import from_limbs
from tests.test_circuits import step_inputs
from vimz_amd import _lib, calldata
from vimz_amd.circuit import Circuit

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from vimz_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


def _limbs64(v):
    return [(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]


def test_groth16_proof_of_the_final_fold_fills_the_calldata_and_the_pairing_check_accepts_it(ctx, oracle):
    from tests import _cyclefold as cfo
    from tests._oracle import T_HASH
    from vimz_amd import hip
    c = Circuit.for_resolution("hash", "HD")
    ck1 = ctx.bases_generate(_lib.CURVE_BN254_G1, 1 << 16)
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-cyclefold")
    z0, inputs = step_inputs("hash")
    cf = hip.CycleFoldIVC(ctx, c, ck1, ck2, max_batch=2)
    dec = None
    try:
        cf.reset(z0); cf.fold(np.stack(inputs[:4]))
        assert cf.verify(4, z0) == 0
        dec = hip.Decider(cf, seed=b"test setup, not a ceremony")
        info = dec.info()
        ci = cf.info()
        # the circuit: the relaxed R1CS of F' + step circuit row by row (one or two constraints a row), two Horner chains, five hashes
        assert info["public_inputs"] == 2 * c.len_z + 2
        assert ci["main_constraints"] + ci["main_wires"] < info["constraints"] < 3 * ci["main_constraints"] + ci["main_wires"] + 20000
        assert info["domain"] >= info["constraints"] + info["public_inputs"] + 1 and info["domain"] & (info["domain"] - 1) == 0
        words, (cW1, cE1), pub = calldata.decider_words(cf, decider=dec)
        assert all(isinstance(w, int) for w in words)
        # the statement the public inputs make: (i, z_0, z_i) and the hash that binds the other words
        z = list(z0)
        for i in range(4):
            ok, z = oracle.step_eval(T_HASH, z, inputs[i])
        lz = c.len_z
        assert pub[0] == 4 and pub[1:1 + lz] == [int(x) for x in z0] and pub[1 + lz:1 + 2 * lz] == z
        dg = cfo.shape_digest(cf)
        pts = [(words[0], words[1]), (words[2], words[3]), (words[4], words[5]), (words[6], words[7]), cW1, cE1]
        # (U_{i+1}'s commitments are what the contract computes itself: U_i.cm + rho·(u_i.cmW | cmT))
        assert cW1 == oracle.curve_add(0, pts[0], oracle.curve_mul(0, pts[2], words[8])) and cE1 == oracle.curve_add(0, pts[1], oracle.curve_mul(0, pts[3], words[8]))
        h_in = [dg, words[8]]
        for x, y in pts:
            h_in += _limbs64(x) + _limbs64(y)
        h_in += [words[17], words[18], words[19], words[20]]
        assert pub[-1] == oracle.nova_hash(0, h_in)
        # Groth16: e(A, B) = e(alpha, beta) · e(sum x_i IC_i, gamma) · e(C, delta)
        vk = dec.verifying_key()
        assert len(vk["ic"]) == len(pub) + 1
        A, C = (words[9], words[10]), (words[15], words[16])
        B = ((words[12], words[11]), (words[14], words[13]))          # the calldata carries the imaginary parts first
        assert bp.g1_on_curve(A) and bp.g1_on_curve(C) and bp.g2_on_curve(B)
        assert bp.groth16_verify(vk, pub, (A, B, C))
        # another statement, another word, another proof element: rejected
        assert not bp.groth16_verify(vk, [pub[0] + 1] + pub[1:], (A, B, C))
        assert not bp.groth16_verify(vk, pub[:-1] + [(pub[-1] + 1) % bp.R], (A, B, C))
        assert not bp.groth16_verify(vk, pub, (A, B, bp.g1_add(C, bp.G1)))
        # the 25 words travel in the reference's calldata layout
        raw = calldata.encode(4, z0, z, words)
        d = calldata.decode(raw)
        assert d["proof"] == words and d["steps"] == 4
        assert cf.verify(4, z0) == 0                         # (the prover is left as it was)
        # the same proof again is the same proof (deterministic randomizers), a further step gives another
        words2, _, pub2 = calldata.decider_words(cf, decider=dec)
        assert words2 == words and pub2 == pub
    finally:
        if dec is not None:
            dec.close()
        cf.close(); ck1.free(); ck2.free()


def test_decider_refuses_a_proof_that_does_not_satisfy_its_statement(ctx, oracle):
    """Wrong KZG evaluations (the e_W the calldata would carry is not p(c_W)): vimz_decider_prove reports UNSAT instead of proving a false statement."""
    from vimz_amd import hip
    c = Circuit.for_resolution("hash", "HD")
    ck1 = ctx.bases_generate(_lib.CURVE_BN254_G1, 1 << 16)
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-cyclefold")
    z0, inputs = step_inputs("hash")
    cf = hip.CycleFoldIVC(ctx, c, ck1, ck2, max_batch=2)
    dec = m = None
    try:
        cf.reset(z0); cf.fold(np.stack(inputs[:2]))
        dec = hip.Decider(cf, seed=b"s")
        m = hip.CycleFoldMerged(cf)
        ev, _ = m.kzg_open(0, 12345)
        ev2, _ = m.kzg_open(1, 6789)
        pub, proof, _ = dec.prove(m, (12345, 6789, ev, ev2))
        assert bp.groth16_verify(dec.verifying_key(), pub, proof)
        with pytest.raises(_lib.VimzError) as e:
            dec.prove(m, (12345, 6789, (ev + 1) % _lib.MODULUS[0], ev2))
        assert e.value.code == _lib.ERR_UNSAT
    finally:
        if m is not None:
            m.close()
        if dec is not None:
            dec.close()
        cf.close(); ck1.free(); ck2.free()
