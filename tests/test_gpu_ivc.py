"""Nova IVC on the GPU (vimz_ivc_*): the product's own verifier must accept, and so must an
independent verifier assembled from the CPU oracle (oracle/nova.hpp hashes, generic relaxed-R1CS check, oracle MSM) over the
exported proof: both running instances, the last fresh secondary instance, the two output hashes and every commitment."""
import hashlib
import struct

import numpy as np
import pytest

from tests._oracle import from_limbs, to_limbs
from tests.test_circuits import step_inputs
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
    ck1 = ctx.bases_generate(_lib.CURVE_BN254_G1, 1 << 19)
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-secondary")
    yield ck1, ck2
    ck1.free(); ck2.free()


def _shape_digest(ivc, side):
    """SHA3-256 of the augmented circuit's shape, as aug/augmented.hpp serialises it, truncated to 250 bits."""
    from vimz_amd import hip
    info = hip._export(ivc.ctx.lib.vimz_ivc_export, ivc.h, side, hip.IX_INFO).view(np.uint64)
    len_z = ivc.circuit.len_z if side == 0 else 1
    h = hashlib.sha3_256()
    h.update(struct.pack("<6Q", 0x3130677561, int(info[0]), int(info[1]), len_z, int(info[2]), 1 if side == 0 else 0))
    for code in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9):     # A,B,C row_ptr/col/coef, then the dictionary in Montgomery form
        a = hip._export(ivc.ctx.lib.vimz_ivc_export, ivc.h, side, code)
        elem = 32 if code == 9 else 4
        h.update(struct.pack("<Q", len(a) // elem))
        h.update(a.tobytes())
    return int.from_bytes(h.digest(), "little") & ((1 << 250) - 1)


def oracle_verify(oracle, ivc, ck1, ck2, n_steps, z0, check_commitments=True):
    """RecursiveSNARK::verify restated over the exported proof; returns a list of failed checks."""
    from vimz_amd import hip
    failed = []
    info = ivc.info()
    par1, par2 = from_limbs(ivc.export(0, hip.IX_PARAMS)), from_limbs(ivc.export(1, hip.IX_PARAMS))
    len_z = info["len_z"]
    digest1, z0_x, z_n = par1[0], par1[1:1 + len_z], par1[1 + len_z:]
    digest2 = par2[0]
    if z0_x != list(z0): failed.append("z0")
    if info["steps"] != n_steps: failed.append("step count")
    # the verifier's OWN digests and the CLAIMED z0 go into the hashes below, never the proof's copies
    digest1, digest2 = _shape_digest(ivc, 0), _shape_digest(ivc, 1)
    if (digest1, digest2) != (par1[0], par2[0]): failed.append("shape digest")
    U1, U2 = from_limbs(ivc.export(0, hip.IX_INSTANCE)), from_limbs(ivc.export(1, hip.IX_INSTANCE))
    u2 = from_limbs(ivc.export(1, hip.IX_FRESH_INSTANCE))
    # the two hashes the last secondary instance carries
    if oracle.nova_instance_hash(0, digest1, n_steps, list(z0), z_n, U2) != u2[2]: failed.append("hash of the primary chain")
    if oracle.nova_instance_hash(1, digest2, n_steps, [0], [0], U1) != u2[3]: failed.append("hash of the secondary chain")
    for side, U, ck, cid, fid in ((0, U1, ck1, 0, 0), (1, U2, ck2, 1, 1)):
        tabs = ivc.r1cs(side)
        Z, E = ivc.export(side, hip.IX_RUNNING_Z), ivc.export(side, hip.IX_RUNNING_E)
        nw = len(Z)
        if from_limbs(Z[0:1])[0] != U[4] or from_limbs(Z[-2:]) != U[5:7]: failed.append(f"side {side}: instance scalars")
        if oracle.r1cs_check_relaxed(fid, tabs, nw, Z, u=U[4], E=E) != -1: failed.append(f"side {side}: relaxed relation")
        if check_commitments:
            bases = ck.download(0, max(nw - 3, len(E)))
            if oracle.msm(cid, bases[:nw - 3], Z[1:nw - 2]) != (U[0], U[1]): failed.append(f"side {side}: comm_W")
            if oracle.msm(cid, bases[:len(E)], E) != (U[2], U[3]): failed.append(f"side {side}: comm_E")
    tabs = ivc.r1cs(1)
    z2 = ivc.export(1, hip.IX_FRESH_Z)
    if from_limbs(z2[0:1])[0] != 1 or from_limbs(z2[-2:]) != u2[2:4]: failed.append("fresh instance scalars")
    if oracle.r1cs_check_relaxed(1, tabs, len(z2), z2) != -1: failed.append("fresh relation")
    if check_commitments:
        bases = ck2.download(0, len(z2) - 3)
        if oracle.msm(1, bases, z2[1:len(z2) - 2]) != (u2[0], u2[1]): failed.append("fresh comm_W")
    return failed, z_n


@pytest.mark.parametrize("op", ["hash", "grayscale", "contrast", "blur", "resize"])
def test_ivc_verifies_and_the_oracle_verifier_accepts(ctx, keys, oracle, op):
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution(op, "HD")
    z0, inputs = step_inputs(op)
    steps = np.stack(inputs)
    ivc = hip.IVC(ctx, c, ck1, ck2, max_batch=4)
    acc = hip.Prover(ctx, c, ck1, max_batch=4)
    try:
        ivc.reset(z0)
        ivc.fold(steps)
        assert ivc.verify(10, z0) == 0
        assert ivc.verify(9, z0) & 4096 and ivc.verify(10, [x + 1 for x in z0]) & (4096 | 1) == (4096 | 1)     # the statement is part of the check
        z_n, n = ivc.state()
        assert n == 10
        acc.reset(z0); acc.fold(steps)
        assert from_limbs(acc.instance()["z"]) == z_n          # the step circuit's state chain is the pinned one
        failed, z_exported = oracle_verify(oracle, ivc, ck1, ck2, 10, z0, check_commitments=op in ("hash", "grayscale"))
        assert failed == [] and z_exported == z_n
        info = ivc.info()
        assert info["step_wires"] == c.n_wires and info["step_constraints"] == c.n_constraints
        assert info["primary_wires"] == c.n_wires + info["verifier_wires"]
    finally:
        ivc.close(); acc.close()


def test_ivc_in_several_calls_equals_one_call_and_reset_restarts(ctx, keys, oracle):
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("hash", "HD")
    z0, inputs = step_inputs("hash")
    steps = np.stack(inputs)
    a, b = hip.IVC(ctx, c, ck1, ck2, max_batch=3), hip.IVC(ctx, c, ck1, ck2, max_batch=16)
    try:
        a.reset(z0); b.reset(z0)
        a.fold(steps[:1]); a.fold(steps[1:4]); a.fold(steps[4:])
        b.fold(steps)
        assert a.verify(10, z0) == 0 and b.verify(10, z0) == 0
        for side in (0, 1):
            assert (a.export(side, hip.IX_INSTANCE) == b.export(side, hip.IX_INSTANCE)).all()
            assert (a.export(side, hip.IX_RUNNING_Z) == b.export(side, hip.IX_RUNNING_Z)).all()
        assert (a.export(1, hip.IX_FRESH_INSTANCE) == b.export(1, hip.IX_FRESH_INSTANCE)).all()
        # verification in the middle of a run does not disturb it
        a.reset(z0); a.fold(steps[:5]); assert a.verify(5, z0) == 0; a.fold(steps[5:]); assert a.verify(10, z0) == 0
        assert (a.export(0, hip.IX_INSTANCE) == b.export(0, hip.IX_INSTANCE)).all()
        failed, _ = oracle_verify(oracle, a, ck1, ck2, 10, z0)
        assert failed == []
    finally:
        a.close(); b.close()


def test_ivc_rejects_a_row_that_violates_the_step_relation(ctx, keys):
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("grayscale", "HD")
    z0, inputs = step_inputs("grayscale")
    steps = np.stack(inputs).copy()
    good = np.stack(inputs)
    steps[6, 200, 0] ^= np.uint64(0xFF)          # a transformed pixel that is no longer the grayscale of the original
    ivc = hip.IVC(ctx, c, ck1, ck2, max_batch=4)
    ref = hip.IVC(ctx, c, ck1, ck2, max_batch=4)
    try:
        ivc.reset(z0)
        with pytest.raises(_lib.VimzError) as e:
            ivc.fold(steps)
        assert e.value.code == _lib.ERR_UNSAT
        # the batch before the bad one (rows 0-3) stays folded and the proof is consistent there: it verifies and can go on
        assert ivc.state()[1] == 4 and ivc.verify(4, z0) == 0
        ivc.fold(good[4:])
        ref.reset(z0); ref.fold(good)
        assert ivc.verify(10, z0) == 0 and ivc.state() == ref.state()
        assert (ivc.export(0, hip.IX_INSTANCE) == ref.export(0, hip.IX_INSTANCE)).all()
    finally:
        ivc.close(); ref.close()


def test_a_forged_instance_is_caught_by_the_oracle_verifier(ctx, keys, oracle):
    """The oracle verifier is not vacuous: flipping one element of the exported proof makes it fail."""
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("hash", "HD")
    z0, inputs = step_inputs("hash")
    ivc = hip.IVC(ctx, c, ck1, ck2, max_batch=4)
    try:
        ivc.reset(z0); ivc.fold(np.stack(inputs)[:4])
        assert oracle_verify(oracle, ivc, ck1, ck2, 4, z0, check_commitments=False)[0] == []
        assert oracle_verify(oracle, ivc, ck1, ck2, 5, z0, check_commitments=False)[0] != []      # wrong step count
        assert "z0" in oracle_verify(oracle, ivc, ck1, ck2, 4, [1], check_commitments=False)[0]
    finally:
        ivc.close()


def test_full_image_as_two_ivc_proofs_ends_in_the_references_committed_state(ctx, keys):
    """All 720 rows of the reference's sample image (img2, contrast 1.4) proven as two Nova IVC proofs of contiguous row
    segments folded concurrently: both verify, the boundary states chain, and the last state is the final state inside the
    reference's committed proof (marketplace/proofs/img2-contrast.proof via tests/golden/kat.json)."""
    from tests import _data
    from vimz_amd import folding, hip, image_editor as ie
    from vimz_amd.distributed import fold_concurrently, ivc_segments
    ck1, ck2 = keys
    P = _data.kat()["proofs"]["img2-contrast"]
    inp = ie.build_input("contrast", _data.load_image("img2"), factor=1.4)
    rows, z0 = folding.prepare_input("contrast", inp, "HD")
    c = Circuit.for_resolution("contrast", "HD")
    ctx2 = hip.Context(0)
    ivcs = [hip.IVC(ctx, c, ck1, ck2, max_batch=32), hip.IVC(ctx2, c, ck1, ck2, max_batch=32)]
    try:
        segs = ivc_segments(ivcs, rows, z0)
        for v, r, z in segs:
            v.reset(z)
        fold_concurrently([(v, r) for v, r, z in segs])
        assert [v.verify(360, z) for v, r, z in segs] == [0, 0]
        assert segs[0][2] == [int(x) for x in z0]
        assert ivcs[0].state() == (segs[1][2], 360)
        assert ivcs[1].state() == ([int(x) for x in P["z_final"]], 360)
    finally:
        for v in ivcs:
            v.close()
        ctx2.close()


def test_proof_export_import_verifies_elsewhere_and_resumes(ctx, keys, oracle):
    """The serialised proof verifies in a fresh vimz_ivc (another prover object = what another process would build), folding
    resumes from it and ends exactly where an uninterrupted run ends; a corrupted blob is rejected by the verifier."""
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("grayscale", "HD")
    z0, inputs = step_inputs("grayscale")
    steps = np.stack(inputs)
    a, b, ref = (hip.IVC(ctx, c, ck1, ck2, max_batch=4) for _ in range(3))
    try:
        a.reset(z0); a.fold(steps[:6])
        blob = a.proof_export()
        b.proof_import(blob)
        assert b.verify(6, z0) == 0 and b.state() == a.state()
        b.fold(steps[6:])
        ref.reset(z0); ref.fold(steps)
        assert b.verify(10, z0) == 0
        for side in (0, 1):
            assert (b.export(side, hip.IX_INSTANCE) == ref.export(side, hip.IX_INSTANCE)).all()
            assert (b.export(side, hip.IX_RUNNING_Z) == ref.export(side, hip.IX_RUNNING_Z)).all()
        assert (b.export(1, hip.IX_FRESH_INSTANCE) == ref.export(1, hip.IX_FRESH_INSTANCE)).all()
        failed, _ = oracle_verify(oracle, b, ck1, ck2, 10, z0, check_commitments=False)
        assert failed == []
        nc2 = a.info()["secondary_constraints"]
        # in the running products / the primary witness / the last fresh secondary witness (the vectors after it — products of the
        # fresh instance and the pending cross term — are resume state that verification recomputes or never reads)
        for where in (len(blob) // 2, 4096, len(blob) - 32 * 8 * nc2 - 40):
            bad = blob.copy()
            bad[where] ^= 1
            a.proof_import(bad)
            assert a.verify(6, z0) != 0, where
        # limbs that are not below the modulus are refused before anything of the importing IVC changes
        b_state = b.state()
        for where in (len(blob) // 2, len(blob) - 16):
            bad = blob.copy()
            bad[where - where % 32: where - where % 32 + 32] = 0xFF
            with pytest.raises(_lib.VimzError):
                b.proof_import(bad)
        assert b.state() == b_state and b.verify(10, z0) == 0
        with pytest.raises(_lib.VimzError):
            a.proof_import(blob[:1000])
    finally:
        a.close(); b.close(); ref.close()


def test_large_msm_split_over_helper_contexts_gives_the_same_proof(ctx, keys):
    """SURVEY.md §8e single-proof multi-GPU: the step's large MSM(T) split by base range over two helper contexts (on a one-GPU box:
    two more contexts of the same device, each with its own stream and workspace; on a node: other devices with key replicas) —
    partial commitments added on the host.  The proof is bit-identical to the unsplit one."""
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("grayscale", "HD")
    z0, inputs = step_inputs("grayscale")
    steps = np.stack(inputs)
    helpers = [hip.Context(0), hip.Context(0)]
    a, b = hip.IVC(ctx, c, ck1, ck2, max_batch=4), hip.IVC(ctx, c, ck1, ck2, max_batch=4)
    try:
        for h in helpers:
            b.add_msm_helper(h, ck1)
        a.reset(z0); a.fold(steps)
        b.reset(z0); b.fold(steps[:4]); b.fold(steps[4:])
        assert a.verify(10, z0) == 0 and b.verify(10, z0) == 0
        for side in (0, 1):
            assert (a.export(side, hip.IX_INSTANCE) == b.export(side, hip.IX_INSTANCE)).all()
        assert (a.export(1, hip.IX_FRESH_INSTANCE) == b.export(1, hip.IX_FRESH_INSTANCE)).all()
    finally:
        a.close(); b.close()
        for h in helpers:
            h.close()


def test_concurrent_provers_all_verify(oracle):
    """Four IVC provers on their own contexts fold concurrently from four host threads, several times over from a fresh state
    (the first step after creation used to be the fragile one: a null-stream fill could land after the first kernel's writes)."""
    from concurrent.futures import ThreadPoolExecutor
    from vimz_amd import hip
    c = Circuit.for_resolution("hash", "HD")
    z0, inputs = step_inputs("hash")
    steps = np.stack(inputs)
    for rep in range(3):
        ctxs = [hip.Context(0) for _ in range(4)]
        ck1 = ctxs[0].bases_generate(_lib.CURVE_BN254_G1, 1 << 14)
        ck2 = ctxs[0].bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-secondary")
        ivcs = [hip.IVC(cx, c, ck1, ck2, max_batch=4) for cx in ctxs]
        try:
            for v in ivcs:
                v.reset(z0)
            with ThreadPoolExecutor(4) as ex:
                list(ex.map(lambda v: v.fold(steps), ivcs))
            assert [v.verify(10, z0) for v in ivcs] == [0, 0, 0, 0]
            ref = ivcs[0].export(0, hip.IX_INSTANCE)
            assert all((v.export(0, hip.IX_INSTANCE) == ref).all() for v in ivcs)      # same inputs: the very same proof
        finally:
            for v in ivcs:
                v.close()
            ck1.free(); ck2.free()
            for cx in ctxs:
                cx.close()


@pytest.mark.gpu
def test_proof_does_not_depend_on_the_schedule():
    """The same rows give the same proof (running instances, fresh instance, state) whichever way the step's work is spread over
    streams, kernels and the host: default (three streams, fused small MSMs over window tables, large MSM queued behind the fused fold, the
    Poseidon jobs of a call's first 24 rows evaluated on host threads) against no / a shorter host-evaluated head batch and the
    debugging switches that serialise or replace each of those pieces (the Poseidon chains' arithmetic among them), and the lookahead schedule (the large MSM's cross term taken one
    step ahead against the previous running instance and completed by the producer's fresh x fresh commitment).  Each variant runs in a process of its own (the switches
    are read once)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    variants = [{}, {"VIMZ_HEAD_ROWS": "0"}, {"VIMZ_HEAD_ROWS": "3"}, {"VIMZ_DEBUG_NO_S2": "1"}, {"VIMZ_DEBUG_NO_SMALL_TABLES": "1"}, {"VIMZ_IVC_MULT_TABLES": "1"}, {"VIMZ_DEBUG_NO_SMALL_MSM": "1"},
                {"VIMZ_DEBUG_SMALL_SUM_KERNEL": "1", "VIMZ_AUG_NO_THREADS": "1"}, {"VIMZ_TUNE": "sort_blocks=256,combine_lane_bits=4"},
                {"VIMZ_IVC_LOOKAHEAD": "1"}, {"VIMZ_TUNE": "sort_blocks=40,combine_lane_bits=2", "VIMZ_DEBUG_CHECK_MSM": "1"},
                # the Poseidon chains in the standard 8 x 32-bit arithmetic instead of the reduced-radix one, with the GPU producing every row
                {"VIMZ_DEBUG_POSEIDON_STD": "1", "VIMZ_HEAD_ROWS": "0"}, {"VIMZ_TUNE": "ones_dense=0"},
                # round 5: the shared bucket set reduced by bit planes, with batches of two rows — the producer's issuer thread then launches MSMs of later rows
                # while the folding thread finishes earlier ones with the SAME plan object (a half-written plan once gave wrong commitments here)
                {"VIMZ_DIGEST_TABLES": "15"}, {"VIMZ_DIGEST_TABLES": "15", "VIMZ_TUNE": "reduce_planes=1", "VIMZ_HEAD_ROWS": "0", "_batch": "2"},
                {"VIMZ_DIGEST_TABLES": "14", "VIMZ_TUNE": "reduce_planes=1"}]
    # (the lookahead also with batches of two and three rows and no host-evaluated head: every way a row two steps ahead can fall into
    #  the same batch, the next one, or not exist)
    variants += [{"VIMZ_IVC_LOOKAHEAD": "1", "VIMZ_HEAD_ROWS": "0", "_batch": "2"}, {"VIMZ_IVC_LOOKAHEAD": "1", "VIMZ_HEAD_ROWS": "0", "_batch": "3"}]
    # round 6: the default commits to the step rows' cross term in its boolean-row form (half the points through the MSM, the rest from a unit-scalar sum and a
    # running commitment kept by linearity); the plain vector, alone and under the lookahead, must give the same commitment — the same proof
    variants += [{"VIMZ_IVC_BOOL_ROWS": "0"}, {"VIMZ_IVC_BOOL_ROWS": "0", "VIMZ_IVC_LOOKAHEAD": "1", "VIMZ_HEAD_ROWS": "0", "_batch": "2"}]
    lines = []
    for env in variants:
        env = dict(env)
        batch = env.pop("_batch", "8")
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "ivc_digest.py"), "grayscale", "2", batch], capture_output=True, text=True,
                             timeout=600, env={**os.environ, **env})
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("digest ")][-1]
        assert " verify 0 " in line, (env, line)
        lines.append(line)
    assert len(set(lines)) == 1, list(zip(variants, lines))
