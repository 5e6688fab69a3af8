"""The decider of the Nova + CycleFold path (vimz_kzg_setup, vimz_decider_*; vimz_amd/csrc/groth16.hip, aug/decider.hpp, pairing.hpp): `Decider::preprocess` /
`prove` / `verify` of the reference's Sonobe backend (vimz/src/sonobe_backend/mod.rs:72-80, decider.rs:13-50).  The checker is tests/_novadecider.py —
the restatement of contracts/*Verifier.sol that the CPU suite pins on the reference's six committed proofs: with THIS library's verifying key as the
contract's constants it must accept THIS library's 25 words, and reject each changed word.  The product's own verifier (vimz_decider_verify) must agree."""
import numpy as np
import pytest

from tests import _novadecider as nd
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


def _prove_and_check(ctx, oracle, op, n_srs, steps, z0, inputs, t_oracle=None, full_negative=False, light=False):
    from vimz_amd import hip
    c = Circuit.for_resolution(op, "HD")
    srs, kzg_vk = hip.kzg_setup(ctx, n_srs)
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-cyclefold")
    cf = hip.CycleFoldIVC(ctx, c, srs, ck2, max_batch=2)
    dec = None
    try:
        cf.reset(z0); cf.fold(np.stack(inputs[:steps]))
        assert cf.verify(steps, z0) == 0
        dec = hip.Decider(cf, kzg_vk=kzg_vk, light=light)
        info, ci = dec.info(), cf.info()
        lz = c.len_z
        assert info["public_inputs"] == 36 + 2 * lz          # the contract's layout: pp_hash, i, z_0, z_i, 4 x 5 limbs, 4 scalars, 2 x 5 limbs
        # the full decider's extra rows do not depend on the step circuit: 2 619 scalars at 3 rows a bit, 1 313 rows of the CycleFold shape at ~570
        assert info["cyclefold_rows"] == 0 if light else 2_700_000 < info["cyclefold_rows"] < 2_800_000
        light_rows = info["constraints"] - info["cyclefold_rows"]
        assert ci["main_constraints"] + ci["main_wires"] < light_rows < 3 * ci["main_constraints"] + ci["main_wires"] + 20000
        assert info["domain"] >= info["constraints"] + info["public_inputs"] + 1 and info["domain"] & (info["domain"] - 1) == 0
        raw, d = calldata.decider_calldata(dec)
        words, pub = d["words"], d["public_inputs"]
        # the statement: (i, z_0, z_i) as the oracle's step function gives them
        if t_oracle is not None:
            z = list(z0)
            for i in range(steps):
                ok, z = oracle.step_eval(t_oracle, z, inputs[i])
            assert d["steps"] == steps and d["z0"] == [int(x) for x in z0] and d["z_i"] == z
        assert len(raw) == 4 + 32 * (1 + 2 * lz + 25) and calldata.decode(raw)["proof"] == words
        # the restated contract, with this library's key as its constants
        key = dec.verifying_key()
        assert key["len_z"] == lz and len(key["groth16"]["ic"]) == len(pub) + 1
        pub_contract, cmW, cmE = nd.public_inputs(key, d["steps"], d["z0"], d["z_i"], words)
        assert pub_contract == pub                             # the prover's public inputs are the ones the contract assembles from the words
        assert words[8] >> 128 == 1                            # r = 2^128 + 128 bits, as F' derives its challenges
        assert nd.verify(key, d["steps"], d["z0"], d["z_i"], words) == (True, "ok")
        assert nd.verify_calldata({"k": key}, "k", raw) == (True, "ok")
        assert dec.verify(d["steps"], d["z0"], d["z_i"], words) == 0
        from vimz_amd.hip import decider_verify_key
        assert decider_verify_key(dec.key_words(), d["steps"], d["z0"], d["z_i"], words) == 0
        # a changed statement or word: rejected by the restated contract and, for the same reason, by the product
        q, r = nd.bp.Q, nd.bp.R
        cases = [("steps", 8), ("z_i", 8), ("eval_W", 2), ("neg_proof_E", 4), ("neg_C", 8), ("r", 6)]
        if full_negative:
            cases += [("z0", 8), ("eval_E", 4), ("challenge_W", 2 | 8), ("neg_cmT", 4 | 8), ("neg_UW", 2 | 8), ("neg_A", 8)]
        for what, bits in cases:
            st, a0, ai, w = d["steps"], list(d["z0"]), list(d["z_i"]), list(words)
            if what == "steps": st += 1
            elif what == "z_i": ai[-1] = (ai[-1] + 1) % r
            elif what == "z0": a0[0] = (a0[0] + 1) % r
            elif what == "eval_W": w[19] = (w[19] + 1) % r
            elif what == "eval_E": w[20] = (w[20] + 1) % r
            elif what == "challenge_W": w[17] = (w[17] + 1) % r
            elif what == "neg_proof_E": w[24] = q - w[24]
            elif what == "neg_C": w[16] = q - w[16]
            elif what == "neg_A": w[10] = q - w[10]
            elif what == "neg_cmT": w[7] = q - w[7]
            elif what == "neg_UW": w[1] = q - w[1]
            elif what == "r": w[8] += 1
            got = dec.verify(st, a0, ai, w)
            ok, why = nd.verify(key, st, a0, ai, w)
            assert not ok and got != 0, what
            # the restated contract stops at its first failing `require`; the product reports every failing check
            first = {"KZG: verifying proof for challenge W": 2, "KZG: verifying proof for challenge E": 4, "Groth16": 8}
            assert any(why.startswith(k) and (got & v) for k, v in first.items()), (what, why, got)
            assert got & bits, (what, got)
        w = list(words); w[0] ^= 1                               # off the curve: the precompile reverts / the product says malformed
        assert dec.verify(d["steps"], d["z0"], d["z_i"], w) & 16 and nd.verify(key, d["steps"], d["z0"], d["z_i"], w)[1].startswith("precompile reverted")
        assert dec.verify(1, d["z0"], d["z_i"], words) & 1
        assert cf.verify(steps, z0) == 0                         # (the prover is left as it was)
        # a second proof of the same statement: other blinding, same public inputs, accepted
        words2, pub2, _ = dec.prove()
        assert pub2 == pub and words2[:9] == words[:9] and words2[17:] == words[17:] and words2[9:17] != words[9:17]
        assert dec.verify(d["steps"], d["z0"], d["z_i"], words2) == 0
        if full_negative:      # the key pair at rest: saved, loaded into a decider of its own (no trapdoor involved), which proves under the same verifying key
            blob = dec.save_key()
            dec3 = hip.Decider.load_key(cf, blob)
            try:
                assert dec3.key_words().tolist() == dec.key_words().tolist() and dec3.info() == info
                words3, pub3, _ = dec3.prove()
                assert pub3 == pub and dec.verify(d["steps"], d["z0"], d["z_i"], words3) == 0 and dec3.verify(d["steps"], d["z0"], d["z_i"], words) == 0
                assert nd.verify(key, d["steps"], d["z0"], d["z_i"], words3) == (True, "ok")
            finally:
                dec3.close()
            for bad in (blob[:-8], np.concatenate([blob[:8] ^ np.uint8(1), blob[8:]])):      # wrong length, wrong magic
                with pytest.raises(_lib.VimzError):
                    hip.Decider.load_key(cf, bad)
            b2 = blob.copy(); b2[8 * 6] ^= 1                                         # another public-parameter hash
            with pytest.raises(_lib.VimzError):
                hip.Decider.load_key(cf, b2)
        return info, d["seconds"], dec.setup_seconds
    finally:
        if dec is not None:
            dec.close()
        cf.close(); srs.free(); ck2.free()


def test_full_decider_words_are_accepted_by_the_restated_contract_hash_step(ctx, oracle):
    """The FULL decider (the reference's default, decider.rs:13-21): same 25 words, same public inputs; the key pair (1.3 GB at rest) saves and loads."""
    from tests._oracle import T_HASH
    z0, inputs = step_inputs("hash")
    info, _, _ = _prove_and_check(ctx, oracle, "hash", 36000, 4, z0, inputs, t_oracle=T_HASH, full_negative=True)
    assert info["domain"] == 1 << 22


def test_light_decider_words_are_accepted_by_the_restated_contract_hash_step(ctx, oracle):
    """The reference's opt-in `light-test` variant (vimz/Cargo.toml:56-59): the circuit of checks 1-4 only."""
    from tests._oracle import T_HASH
    z0, inputs = step_inputs("hash")
    _prove_and_check(ctx, oracle, "hash", 36000, 4, z0, inputs, t_oracle=T_HASH, full_negative=True, light=True)


def test_full_decider_at_contrast_hd(ctx, oracle):
    """BASELINE.json's headline circuit: contrast HD — 1.04 M rows for the folded main instance, the hashes and the KZG evaluations + 2.75 M for the
    CycleFold instance: domain 2^22, SRS of 2^19 powers made on the GPU."""
    from tests._oracle import T_CONTRAST
    z0, inputs = step_inputs("contrast")
    info, sec, setup = _prove_and_check(ctx, oracle, "contrast", 1 << 19, 3, z0, inputs, t_oracle=T_CONTRAST)
    assert 3_700_000 < info["constraints"] < (1 << 22) - 64 and info["domain"] == 1 << 22


def test_light_decider_at_contrast_hd(ctx, oracle):
    from tests._oracle import T_CONTRAST
    z0, inputs = step_inputs("contrast")
    info, sec, setup = _prove_and_check(ctx, oracle, "contrast", 1 << 19, 3, z0, inputs, t_oracle=T_CONTRAST, light=True)
    assert info["constraints"] > 1_000_000 and info["domain"] == 1 << 20


@pytest.mark.parametrize("op,n_srs,light", [("grayscale", 1 << 18, True), ("blur", 1 << 19, False)])
def test_decider_at_the_other_state_widths(ctx, oracle, op, n_srs, light):
    """The reference's nine contracts come in four state widths (tests/golden/verifier_keys.json: len_z 1, 2, 3, 4 — 38, 40, 42, 44 public inputs);
    hash and contrast above are 1 and 3, these are 2 (grayscale, as Grayscale/Redact/ResizeVerifier.sol; light) and 4 (blur, as Blur/SharpnessVerifier.sol; full)."""
    from tests.test_circuits import ORC_T
    z0, inputs = step_inputs(op)
    info, _, _ = _prove_and_check(ctx, oracle, op, n_srs, 3, z0, inputs, t_oracle=ORC_T[op], light=light)
    assert info["public_inputs"] == {"grayscale": 40, "blur": 44}[op] == len(nd.verifier_keys()[op]["groth16"]["ic"]) - 1


def test_decider_refuses_other_shapes_and_a_prover_without_steps(ctx):
    from vimz_amd import hip
    c = Circuit.for_resolution("hash", "HD")
    srs, kzg_vk = hip.kzg_setup(ctx, 36000)
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-cyclefold")
    z0, inputs = step_inputs("hash")
    cf = hip.CycleFoldIVC(ctx, c, srs, ck2, max_batch=2)
    dec = nokey = None
    try:
        cf.reset(z0)
        dec = hip.Decider(cf, kzg_vk=kzg_vk, light=True)
        with pytest.raises(_lib.VimzError):
            dec.prove()                                          # no steps
        nokey = hip.Decider(cf, light=True)
        cf.fold(np.stack(inputs[:2]))
        words, pub, _ = nokey.prove()
        with pytest.raises(_lib.VimzError):
            nokey.verify(2, pub[2:3], pub[3:4], words)           # set up without the SRS's verifying key
        bad = np.array(kzg_vk); bad[0, 0] ^= 1
        with pytest.raises(_lib.VimzError):
            hip.Decider(cf, kzg_vk=bad, light=True)              # not a point of G2
        srs2, vk2 = hip.kzg_setup(ctx, 16)
        srs2.free()
        with pytest.raises(_lib.VimzError, match="not .tau.G2 of the SRS"):
            hip.Decider(cf, kzg_vk=vk2, light=True)              # [tau']G2 of another SRS than the prover commits with
    finally:
        for o in (dec, nokey):
            if o is not None:
                o.close()
        cf.close(); srs.free(); ck2.free()


def test_what_the_full_and_the_light_decider_attest():
    """The FULL decider refuses (VIMZ_ERR_UNSAT) a CycleFold witness or error vector that violates its relation — the instance is attested inside the
    circuit, as Sonobe's `DeciderEth` does; the LIGHT variant binds it by hash only, so there the contract's checks still pass and only the full IVC
    verifier objects; a violated MAIN relation is refused by both.  Also: seeded setups are reproducible.  Body: tests/_tamper_decider.py on
    libvimz_hip_testing.so (vimz_cf_poke, seeded setups)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VIMZ_HIP_LIBRARY="testing")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_tamper_decider.py")], env=env, capture_output=True, text=True, timeout=1500, cwd=root)
    assert r.returncode == 0 and "tamper ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
