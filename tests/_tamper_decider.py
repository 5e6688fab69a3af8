"""Body of tests/test_gpu_decider.py::test_what_the_decider_proof_does_not_attest, run as a process of its own with VIMZ_HIP_LIBRARY=testing (vimz_cf_poke and the
seeded setups exist only in libvimz_hip_testing.so).  Test infrastructure.

The FULL decider (round 6; `DeciderEth`, vimz/src/sonobe_backend/decider.rs:13-21) attests the running CycleFold instance cfU_i inside the circuit: a CycleFold
WITNESS that does not satisfy its relaxed relation — or does not open cfU_i's commitments — gets no proof (VIMZ_ERR_UNSAT), like a wrong MAIN witness.  The LIGHT
variant (the reference's opt-in `light-test` feature, vimz/Cargo.toml:56-59) binds cfU_i by its hash only: there such a witness is invisible to the 25 calldata
words — the contract's checks (vimz_decider_verify, tests/_novadecider.py) still pass and only the full IVC verification (vimz_cf_verify) rejects.  Both recorded."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from tests import _novadecider as nd
    from tests._oracle import from_limbs
    from tests.test_circuits import step_inputs
    from vimz_amd import _lib, hip
    from vimz_amd.circuit import Circuit
    assert _lib.SO_PATH == _lib.TESTING_SO_PATH, "start this script with VIMZ_HIP_LIBRARY=testing"
    ctx = hip.Context(0)
    srs, kzg_vk = hip.kzg_setup(ctx, 36000, seed=b"tamper-decider srs")
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-cyclefold")
    c = Circuit.for_resolution("hash", "HD")
    z0, inputs = step_inputs("hash")
    cf = hip.CycleFoldIVC(ctx, c, srs, ck2, max_batch=2)
    dec = dec2 = light = None
    try:
        cf.reset(z0); cf.fold(np.stack(inputs[:3]))
        assert cf.verify(3, z0) == 0
        dec = hip.Decider(cf, kzg_vk=kzg_vk, seed=b"tamper-decider key")
        words, pub, _ = dec.prove()
        lz = c.len_z
        steps, a0, ai = pub[1], pub[2:2 + lz], pub[2 + lz:2 + 2 * lz]
        assert dec.verify(steps, a0, ai, words) == 0
        # seeded setups are reproducible: the same seed gives the same verifying key (and another seed another one)
        dec2 = hip.Decider(cf, kzg_vk=kzg_vk, seed=b"tamper-decider key")
        assert dec2.key_words().tolist() == dec.key_words().tolist()
        dec2.close(); dec2 = hip.Decider(cf, kzg_vk=kzg_vk, seed=b"another key")
        assert dec2.key_words().tolist() != dec.key_words().tolist() and dec2.verify(steps, a0, ai, words) == 8      # another key: Groth16 fails, KZG passes
        assert dec.info()["cyclefold_rows"] > 2_700_000
        light = hip.Decider(cf, kzg_vk=kzg_vk, seed=b"tamper-decider key", light=True)
        assert light.info()["cyclefold_rows"] == 0 and light.info()["public_inputs"] == dec.info()["public_inputs"]
        words_l, pub_l, _ = light.prove()
        assert pub_l == pub and light.verify(steps, a0, ai, words_l) == 0 and dec.verify(steps, a0, ai, words_l) == 8      # (another circuit: another key)
        # (1) a CycleFold witness that no longer satisfies its relation: the IVC verifier rejects; the FULL decider refuses to prove; the LIGHT one's words do not notice
        vec = from_limbs(cf.export(1, hip.IX_RUNNING_Z))
        idx = 5
        cf.poke(2, idx, (vec[idx] + 1) % _lib.MODULUS[1])
        assert cf.verify(3, z0) != 0
        try:
            dec.prove()
            raise AssertionError("the full decider proved over a CycleFold witness that violates its relation")
        except _lib.VimzError as e:
            assert e.code == _lib.ERR_UNSAT, e
        words_b, pub_b, _ = light.prove()
        assert pub_b == pub and light.verify(steps, a0, ai, words_b) == 0
        assert nd.verify(light.verifying_key(), steps, a0, ai, words_b) == (True, "ok")
        cf.poke(2, idx, vec[idx])
        assert cf.verify(3, z0) == 0
        # ... and a changed element of its error vector likewise
        evec = from_limbs(cf.export(1, hip.IX_RUNNING_E))
        cf.poke(4, 7, (evec[7] + 1) % _lib.MODULUS[1])
        assert cf.verify(3, z0) != 0
        try:
            dec.prove()
            raise AssertionError("the full decider proved over a CycleFold error vector that violates its relation")
        except _lib.VimzError as e:
            assert e.code == _lib.ERR_UNSAT, e
        cf.poke(4, 7, evec[7])
        assert cf.verify(3, z0) == 0
        # (2) a main witness that no longer satisfies the relaxed relation: no proof
        vec = from_limbs(cf.export(0, hip.IX_RUNNING_Z))
        idx = 7
        cf.poke(0, idx, (vec[idx] + 1) % _lib.MODULUS[0])
        try:
            dec.prove()
            raise AssertionError("the decider proved a false relaxed relation")
        except _lib.VimzError as e:
            assert e.code == _lib.ERR_UNSAT, e
        cf.poke(0, idx, vec[idx])
        words_c, pub_c, _ = dec.prove()
        assert pub_c == pub and dec.verify(steps, a0, ai, words_c) == 0
    finally:
        for o in (dec, dec2, light):
            if o is not None:
                o.close()
        cf.close(); srs.free(); ck2.free(); ctx.close()
    print("tamper ok")


if __name__ == "__main__":
    main()
