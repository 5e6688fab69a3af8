"""CompressedSNARK on the GPU (vimz_ivc_compress / vimz_ivc_verify_compressed; SURVEY.md §8f row N2, reference call sites
vimz/src/nova_snark_backend/mod.rs:52-67): the product's verifier accepts, the independent Python verifier of tests/_spartan.py
accepts, both reject for another statement and for tampered proofs."""
import numpy as np
import pytest

from tests import _spartan
from tests.test_circuits import step_inputs
from tests.test_gpu_ivc import _shape_digest
from vimz_amd import _lib
from vimz_amd.circuit import Circuit

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from vimz_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def keys(ctx):
    ck1 = ctx.bases_generate(_lib.CURVE_BN254_G1, 1 << 15)
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-secondary")
    yield ck1, ck2
    ck1.free(); ck2.free()


def test_compressed_proof_of_a_ten_step_hash_ivc(ctx, keys, oracle):
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("hash", "HD")
    z0, inputs = step_inputs("hash")
    ivc = hip.IVC(ctx, c, ck1, ck2, max_batch=4)
    vk = hip.IVC(ctx, c, ck1, ck2, max_batch=1)            # another object = what a verifying process would construct
    try:
        ivc.reset(z0); ivc.fold(np.stack(inputs))
        assert ivc.verify(10, z0) == 0
        proof, t = ivc.compress()
        assert len(proof) < 64 * 1024                       # a few tens of kilobytes whatever the number of steps
        assert vk.verify_compressed(proof, 10, z0) == 0
        assert vk.verify_compressed(proof, 9, z0) & 4096
        assert vk.verify_compressed(proof, 10, [z0[0] + 1]) & (4096 | 1) == (4096 | 1)
        # the independent verifier
        info = ivc.info()
        n1, n2 = (info["primary_wires"], info["primary_constraints"]), (info["secondary_wires"], info["secondary_constraints"])
        key1, key2 = ck1.download(0, 1 << 15), ck2.download(0, 1 << 13)
        args = (_shape_digest(ivc, 0), _shape_digest(ivc, 1), ivc.r1cs(0), ivc.r1cs(1), n1, n2, key1, key2)
        failed, zn = _spartan.verify_compressed(oracle, proof, 10, z0, *args)
        assert failed == [] and zn == ivc.state()[0]
        assert _spartan.verify_compressed(oracle, proof, 11, z0, *args)[0] != []
        # tampering: a sum-check message, a claimed evaluation, an IPA point, the final scalar, an instance coordinate
        words = len(proof) // 8
        for where in (8 * 40, 8 * (words // 3), 8 * (words // 2), len(proof) - 8 * 3, 8 * 12):
            bad = proof.copy()
            bad[where] ^= 1
            assert vk.verify_compressed(bad, 10, z0) != 0, where
        bad = proof.copy()
        bad[8 * (words // 3)] ^= 1
        assert _spartan.verify_compressed(oracle, bad, 10, z0, *args)[0] != []
        # folding on and compressing again gives a proof for the longer statement; the old one stays valid for its own
        ivc.fold(np.stack(inputs)[:3])
        proof2, _ = ivc.compress()
        assert vk.verify_compressed(proof2, 13, z0) == 0 and vk.verify_compressed(proof, 10, z0) == 0
        assert vk.verify_compressed(proof2, 10, z0) != 0
    finally:
        ivc.close(); vk.close()


def test_a_correction_hidden_in_a_public_slot_is_rejected():
    """the W opening must not let a prover hide a correction of a public entry under the (live, otherwise unused)
    generator of that entry's slot.  A cheating prover — the test hook vimz_test_forge_public_slot, which exists only in libvimz_hip_testing.so — claims
    x0 + 1 for the last fresh instance, commits to W − ck[n−3] and opens a vector with −1 in the slot of wire n−2: every sum-check
    message is honest for the TRUE z.  Before the mask on b this argument was accepted (bit 4 clear); it must fail.  The hook is
    switched per call, so the honest proof of the same process is the control."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import numpy as np, os, sys
sys.path.insert(0, %r)
from tests.test_circuits import step_inputs
from vimz_amd import _lib, hip
from vimz_amd.circuit import Circuit
ctx = hip.Context(0)
ck1 = ctx.bases_generate(_lib.CURVE_BN254_G1, 1 << 15)
ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-secondary")
c = Circuit.for_resolution("hash", "HD")
z0, inputs = step_inputs("hash")
ivc = hip.IVC(ctx, c, ck1, ck2, max_batch=4)
ivc.reset(z0); ivc.fold(np.stack(inputs)[:3])
honest, _ = ivc.compress()
assert _lib.SO_PATH == _lib.TESTING_SO_PATH
ctx.lib.vimz_test_forge_public_slot.restype = None
ctx.lib.vimz_test_forge_public_slot(1)
forged, _ = ivc.compress()
ctx.lib.vimz_test_forge_public_slot(0)
print("codes", ivc.verify_compressed(honest, 3, z0), ivc.verify_compressed(forged, 3, z0))
""" % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, VIMZ_HIP_LIBRARY="testing"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    honest, forged = [int(x) for x in [l for l in out.stdout.splitlines() if l.startswith("codes ")][-1].split()[1:]]
    assert honest == 0
    assert forged & 16, forged        # the argument for the last fresh secondary instance itself fails (not only the chain hash, bit 0)


def test_compressed_proof_through_the_reference_call_sequence(ctx, oracle):
    """prepare_folding -> fold_input -> verify_folded_proof -> compress -> verify, as nova_snark_backend::run does (mod.rs:22-67),
    on the grayscale circuit (2^17 rows)."""
    from vimz_amd import folding
    circuit, params = folding.prepare_folding(ctx, "grayscale", "HD")
    z0, inputs = step_inputs("grayscale")
    try:
        proof = folding.fold_input(params, np.stack(inputs[:4]), z0, max_batch=4)
        folding.verify_folded_proof(proof, params, 4, z0)
        blob, t = folding.compress_proof(params, proof)
        folding.verify_compressed_proof(proof.prover, blob, 4, z0)
        with pytest.raises(_lib.VimzError):
            folding.verify_compressed_proof(proof.prover, blob, 5, z0)
        assert t["prove_s"] < 5.0
        proof.prover.close()
    finally:
        params.free()
