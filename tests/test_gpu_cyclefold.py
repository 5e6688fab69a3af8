"""Nova + CycleFold IVC on the GPU (vimz_cf_*, SURVEY.md §8 row N1: the prove_step loop of the reference's Sonobe backend,
vimz/src/sonobe_backend/folding.rs:52-66).  The product's verifier must accept, and so must the independent verifier assembled from the
CPU oracle (tests/_cyclefold.py) over the exported proof; states must equal the Nova IVC's and the oracle executor's."""
import numpy as np
import pytest

from tests import _cyclefold as cfo
from tests._oracle import from_limbs
from tests.test_circuits import ORC_T, step_inputs
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
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-cyclefold")
    yield ck1, ck2
    ck1.free(); ck2.free()


@pytest.mark.parametrize("op", ["contrast", "grayscale", "hash"])
def test_cyclefold_ivc_verifies_and_the_oracle_verifier_accepts(ctx, keys, oracle, op):
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution(op, "HD")
    z0, inputs = step_inputs(op)
    steps = np.stack(inputs)
    cf = hip.CycleFoldIVC(ctx, c, ck1, ck2, max_batch=4)
    ivc = hip.IVC(ctx, c, ck1, ck2, max_batch=4)
    try:
        cf.reset(z0)
        cf.fold(steps)                                  # 10 steps: three batches (4, 4, 2)
        assert cf.verify(10, z0) == 0
        assert cf.verify(9, z0) == 4096 and cf.verify(10, [z0[0] + 1] + list(z0[1:])) & 4096
        failed, z_n = cfo.verify(oracle, cf, ck1, ck2, 10, z0, check_commitments=(op == "contrast"))
        assert failed == []
        # the same ten rows as a Nova IVC end in the same state
        ivc.reset(z0); ivc.fold(steps)
        assert cf.state() == ivc.state() and cf.state()[0] == z_n
        # every CycleFold instance the running one absorbed spoke about a true statement: the running instance's public elements are
        # random combinations — spot-check the relation the circuit enforces on a fresh run of one step instead (below); here: the
        # proof stays valid when folding goes on in a second call
        cf.fold(steps[:3])
        assert cf.verify(13, z0) == 0
        failed, _ = cfo.verify(oracle, cf, ck1, ck2, 13, z0, check_commitments=False)
        assert failed == []
    finally:
        cf.close(); ivc.close()


def test_cyclefold_verifiers_reject_tampering():
    """One wrong element in any of the five witness vectors is rejected by the product's verifier and by the oracle-side one; an
    unsatisfiable row is refused and leaves the proof valid.  Overwriting a prover's vectors needs the test hook vimz_cf_poke, which the
    product library does not have: the body (tests/_tamper_cyclefold.py) runs in a process of its own on libvimz_hip_testing.so."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VIMZ_HIP_LIBRARY="testing")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_tamper_cyclefold.py")], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and "tamper ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_cyclefold_instance_statement(oracle):
    """The statement a CycleFold instance makes, on the host circuit: P3 = P1 + r·P2 for the public elements the witness ends in."""
    g = (1, 2)
    p1, p2 = oracle.curve_mul(0, g, 12345), oracle.curve_mul(0, g, 67890)
    r = (1 << 128) + 0x1234567890abcdef1234567890abcdef
    assert cfo.cyclefold_relation(oracle, [r, p1[0], p1[1], p2[0], p2[1], *oracle.curve_add(0, p1, oracle.curve_mul(0, p2, r))])


def test_fold_input_with_the_sonobe_backend(ctx, oracle):
    """folding.fold_input(mode="cyclefold") / verify_folded_proof: the Sonobe backend's fold_input and verify_folding
    (vimz/src/sonobe_backend/folding.rs:52-75) over ten rows of the sample image; the final state is the image-hash pair the reference checks
    (verify_final_state_arkworks, folding.rs:77-131)."""
    from vimz_amd import folding
    z0, inputs = step_inputs("grayscale")
    steps = np.stack(inputs)
    circuit, params = folding.prepare_folding(ctx, "grayscale", "HD", backend="sonobe")
    try:
        proof = folding.fold_input(params, steps, z0, mode="cyclefold")
        folding.verify_folded_proof(proof, params, 10, z0)
        with pytest.raises(_lib.VimzError):
            folding.verify_folded_proof(proof, params, 9, z0)
        z = list(z0)
        for i in range(10):
            ok, z = oracle.step_eval(ORC_T["grayscale"], z, steps[i])
            assert ok
        assert proof.state() == z
        proof.close()
    finally:
        params.free()


def test_cyclefold_proof_export_import_and_resume(ctx, keys):
    """The proof as bytes (Sonobe's ivc_proof / from_ivc_proof): another prover object verifies it and keeps folding; the result equals
    the uninterrupted run; malformed blobs are refused and leave the importer untouched."""
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("contrast", "HD")
    z0, inputs = step_inputs("contrast")
    steps = np.stack(inputs)
    a = hip.CycleFoldIVC(ctx, c, ck1, ck2, max_batch=4)
    b = hip.CycleFoldIVC(ctx, c, ck1, ck2, max_batch=4)
    try:
        a.reset(z0); a.fold(steps[:6])
        blob = a.proof_export()
        b.reset(z0)
        b.proof_import(blob)
        assert b.verify(6, z0) == 0 and b.state() == a.state()
        a.fold(steps[6:]); b.fold(steps[6:])
        assert a.verify(10, z0) == 0 and b.verify(10, z0) == 0
        for side in (0, 1):
            assert (a.export(side, hip.IX_INSTANCE) == b.export(side, hip.IX_INSTANCE)).all()
        assert (a.export(0, hip.IX_FRESH_INSTANCE) == b.export(0, hip.IX_FRESH_INSTANCE)).all()
        # refused: truncated, wrong magic, an element above the modulus, a point off its curve
        for bad in (blob[:1000], np.concatenate([np.frombuffer(b"\x00" * 8, dtype=np.uint8), blob[8:]])):
            with pytest.raises(_lib.VimzError):
                b.proof_import(bad)
        t = blob.copy(); t[-32:] = 0xFF
        with pytest.raises(_lib.VimzError):
            b.proof_import(t)
        t = blob.copy(); t[64 + 8 * 4 * 4 + 8 * 4 * 3] ^= 1          # (inside the header's host state: a coordinate of a commitment)
        try:
            b.proof_import(t)
            assert b.verify(6, z0) != 0                               # a change the range / curve checks cannot see is caught by verification
            b.proof_import(blob)
        except _lib.VimzError:
            pass
        assert b.verify(10, z0) == 0 or b.verify(6, z0) == 0          # the importer still holds a valid proof
    finally:
        a.close(); b.close()


def test_merged_cyclefold_proof_of_three_segments(ctx, keys, oracle):
    """ONE proof object out of three row segments' CycleFold proofs (vimz_cf_merge): the product verifier and the oracle-side replay accept
    it for (all rows, z0) and for nothing else; segments in the wrong order are refused; a wrong vector element is rejected."""
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("contrast", "HD")
    z0, inputs = step_inputs("contrast")
    steps = np.stack(inputs)
    ctxs = [ctx, hip.Context(0), hip.Context(0)]
    cfs = [hip.CycleFoldIVC(cx, c, ck1, ck2, max_batch=4) for cx in ctxs]
    m = None
    try:
        bounds = [(0, 4), (4, 7), (7, 10)]
        z = list(z0)
        for v, (lo, hi) in zip(cfs, bounds):
            v.reset(z)
            zs = v.state_chain(z, steps[lo:hi])
            v.fold(steps[lo:hi])
            z = [sum(int(a[k]) << (64 * k) for k in range(4)) for a in zs[-1]]
            assert v.state()[0] == z and v.verify(hi - lo, v_z0(v)) == 0
        with pytest.raises(_lib.VimzError):
            hip.CycleFoldMerged.of([cfs[0], cfs[2]])                 # not adjacent
        m = hip.CycleFoldMerged.of(cfs)
        assert m.info()["segments"] == 3 and m.info()["steps"] == 10
        assert m.verify(10, z0) == 0
        assert m.verify(9, z0) & 4096 and m.verify(10, [z0[0] + 1] + list(z0[1:])) & 4096
        assert m.state() == (list(z0), z, 10)
        failed, acc = cfo.verify_merged(oracle, m, cfs[0], ck1, ck2, 10, z0)
        assert failed == []
        # the same rows as one chain end in the same state
        one = hip.CycleFoldIVC(ctx, c, ck1, ck2, max_batch=4)
        try:
            one.reset(z0); one.fold(steps)
            assert one.state()[0] == z
        finally:
            one.close()
    finally:
        if m is not None:
            m.close()
        for v in cfs:
            v.close()
        for cx in ctxs[1:]:
            cx.close()


def v_z0(v):
    """the initial state a prover was reset to (its proof's own statement)"""
    from vimz_amd import hip
    par = from_limbs(v.export(0, hip.IX_PARAMS))
    return par[1:1 + v.circuit.len_z]


def test_fold_input_cyclefold_in_segments(ctx, oracle):
    from vimz_amd import folding
    z0, inputs = step_inputs("grayscale")
    steps = np.stack(inputs)
    circuit, params = folding.prepare_folding(ctx, "grayscale", "HD", backend="sonobe")
    try:
        proof = folding.fold_input(params, steps, z0, mode="cyclefold", segments=3)
        assert proof.mode == "cyclefold-merged"
        folding.verify_folded_proof(proof, params, 10, z0)
        with pytest.raises(_lib.VimzError):
            folding.verify_folded_proof(proof, params, 10, [1] + list(z0[1:]))
        z = list(z0)
        for i in range(10):
            ok, z = oracle.step_eval(ORC_T["grayscale"], z, steps[i])
        assert proof.state() == z
        proof.close()
    finally:
        params.free()


def test_merged_cyclefold_proof_outlives_nothing_it_should_not(ctx, keys):
    """Freeing the verifier-key prover first orphans the merged proof: later calls fail cleanly, freeing it stays safe."""
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("hash", "HD")
    z0, inputs = step_inputs("hash")
    v = hip.CycleFoldIVC(ctx, c, ck1, ck2, max_batch=2)
    v.reset(z0); v.fold(np.stack(inputs[:2]))
    m = hip.CycleFoldMerged(v)
    assert m.verify(2, z0) == 0
    v.close()
    with pytest.raises(_lib.VimzError):
        m.verify(2, z0)
    with pytest.raises(_lib.VimzError):
        m.info()
    m.close()


def test_kzg_openings_of_a_cyclefold_proof_over_an_srs(ctx, oracle):
    """The Sonobe backend commits the main instances with KZG (folding.rs:22) and its decider opens the final commitments (decider.rs:13-21).
    With the powers [tau^i]G of a test SRS (tau known) uploaded as ck_main, a CycleFold IVC folds and verifies as with any key, and the KZG
    openings of its running comm_W / comm_E (vimz_cf_kzg_open) satisfy the pairing equation's G1 form (tau - z)·proof == comm - eval·G."""
    from tests._oracle import to_limbs
    from vimz_amd import hip
    r = _lib.MODULUS[0]
    c = Circuit.for_resolution("hash", "HD")
    n = 36000                                               # >= max(wires, constraints) of F + F' for this step circuit (34.3 k)
    tau = 0x2718281828459045235360287471352662497757247093699959574966967627 % r
    G = (1, 2)
    srs, t = np.zeros((n, 8), dtype=np.uint64), 1
    for i in range(n):
        srs[i] = to_limbs(list(oracle.curve_mul(0, G, t))).reshape(-1)
        t = t * tau % r
    B = ctx.bases_upload(0, srs)
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-cyclefold")
    z0, inputs = step_inputs("hash")
    cf = hip.CycleFoldIVC(ctx, c, B, ck2, max_batch=2)
    try:
        cf.reset(z0); cf.fold(np.stack(inputs[:4]))
        assert cf.verify(4, z0) == 0
        U = from_limbs(cf.export(0, hip.IX_INSTANCE))
        rs = np.random.default_rng(11)
        for which, comm in ((0, (U[0], U[1])), (1, (U[2], U[3]))):
            for z in (3, int(rs.integers(1, 1 << 62)) * int(rs.integers(1, 1 << 62)) % r):
                ev, proof = cf.kzg_open(which, z)
                lhs = oracle.curve_mul(0, proof, (tau - z) % r)
                rhs = oracle.curve_add(0, comm, oracle.curve_mul(0, G, (r - ev) % r))
                assert lhs == rhs, (which, z)
        assert cf.verify(4, z0) == 0                         # (the prover is left as it was)
    finally:
        cf.close(); B.free(); ck2.free()


def test_merged_cyclefold_runs_save_load_and_fold_into_one_object(ctx, keys, oracle):
    """Two merged objects — one run of segments each, as two GPUs of a sharded proof would make them — travel as bytes, are loaded into
    another prover's context and folded into ONE object (vimz_cf_merge_merged): product verifier and oracle-side replay accept it for
    (all rows, z0) only; runs in the wrong order and tampered blobs are refused."""
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("contrast", "HD")
    z0, inputs = step_inputs("contrast")
    steps = np.stack(inputs)
    ctxs = [ctx, hip.Context(0), hip.Context(0)]
    cfs = [hip.CycleFoldIVC(cx, c, ck1, ck2, max_batch=4) for cx in ctxs]
    objs = []
    try:
        bounds = [(0, 3), (3, 6), (6, 10)]
        z = list(z0)
        for v, (lo, hi) in zip(cfs, bounds):
            v.reset(z); v.fold(steps[lo:hi])
            z = v.state()[0]
        a = hip.CycleFoldMerged.of(cfs[:2]); objs.append(a)          # run A: segments 0, 1
        b = hip.CycleFoldMerged(cfs[2]); objs.append(b)             # run B: segment 2
        blob_a, blob_b = a.save(), b.save()
        # loaded elsewhere (here: the third prover's context supplies shapes and keys)
        la = hip.CycleFoldMerged.load(cfs[2], blob_a); objs.append(la)
        lb = hip.CycleFoldMerged.load(cfs[2], blob_b); objs.append(lb)
        assert la.verify(6, z0) == 0 and la.state()[2] == 6
        with pytest.raises(_lib.VimzError):
            lb.merge(la)                                             # wrong order: B does not end where A starts
        la.merge(lb)
        assert la.info()["segments"] == 3 and la.verify(10, z0) == 0 and la.verify(9, z0) & 4096
        assert la.state() == (list(z0), z, 10)
        failed, acc = cfo.verify_merged(oracle, la, cfs[2], ck1, ck2, 10, z0, check_commitments=False)
        assert failed == []
        with pytest.raises(_lib.VimzError):
            la.merge(cfs[0])                                         # an object of several runs takes no more single segments
        # the whole thing once more through bytes
        again = hip.CycleFoldMerged.load(cfs[0], la.save()); objs.append(again)
        assert again.verify(10, z0) == 0
        # refused or rejected: truncated, an element above the modulus, a changed statement word
        for bad in (blob_a[:4096], ):
            with pytest.raises(_lib.VimzError):
                hip.CycleFoldMerged.load(cfs[2], bad)
        t = blob_a.copy(); t[-32:] = 0xFF
        with pytest.raises(_lib.VimzError):
            hip.CycleFoldMerged.load(cfs[2], t)
        t = blob_a.copy(); t[8 * (8 + 1)] ^= 1                       # n of the first segment
        try:
            x = hip.CycleFoldMerged.load(cfs[2], t); objs.append(x)
            assert x.verify(6, z0) != 0 and x.verify(7, z0) != 0
        except _lib.VimzError:
            pass
    finally:
        for o in objs:
            o.close()
        for v in cfs:
            v.close()
        for cx in ctxs[1:]:
            cx.close()


def test_full_image_with_the_sonobe_scheme_ends_in_the_references_committed_state(ctx):
    """All 720 rows of the reference's sample image (img2, contrast 1.4) through fold_input(mode="cyclefold", segments=3): ONE merged Nova +
    CycleFold proof that verifies for (720 steps, the transformation's z0) and whose final state is the one inside the proof the reference
    committed for this image — which its Sonobe backend produced (marketplace/proofs/img2-contrast.proof via tests/golden/kat.json)."""
    from tests import _data
    from vimz_amd import folding, image_editor as ie
    P = _data.kat()["proofs"]["img2-contrast"]
    inp = ie.build_input("contrast", _data.load_image("img2"), factor=1.4)
    rows, z0 = folding.prepare_input("contrast", inp, "HD")
    circuit, params = folding.prepare_folding(ctx, "contrast", "HD", backend="sonobe")
    try:
        proof = folding.fold_input(params, rows, z0, max_batch=32, mode="cyclefold", segments=3)
        try:
            folding.verify_folded_proof(proof, params, 720, z0)
            assert proof.state() == [int(x) for x in P["z_final"]]
            with pytest.raises(_lib.VimzError):
                folding.verify_folded_proof(proof, params, 719, z0)
        finally:
            proof.close()
    finally:
        params.free()


@pytest.mark.parametrize("n_steps", [1, 2, 5])
def test_step_relation_restated_natively(ctx, keys, oracle, n_steps):
    """What the last step's F' returned equals the native restatement of its relation (tests/_cyclefold.py::step_relation: oracle Poseidon,
    Python integers, oracle curve arithmetic on both curves) applied to what it was given — the base case (1 step), the first fold into
    the zero instances (2) and a general step (5)."""
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("grayscale", "HD")
    z0, inputs = step_inputs("grayscale")
    cf = hip.CycleFoldIVC(ctx, c, ck1, ck2, max_batch=4)
    try:
        cf.reset(z0); cf.fold(np.stack(inputs[:n_steps]))
        assert cf.verify(n_steps, z0) == 0
        s = cfo.parse_last_step(cf.export(0, hip.IX_LAST_STEP), c.len_z)
        dg = cfo.shape_digest(cf)
        assert s["i"] == n_steps - 1
        failed, Un, cn, x0, x1, (r, r1, r2) = cfo.step_relation(oracle, dg, list(z0), s)
        assert failed == []
        assert (Un, cn, x0, x1) == (s["U_new"], s["cfU_new"], s["x0"], s["x1"])
        mask = (1 << 128) - 1
        assert (r & mask, r1 & mask, r2 & mask) == (s["r"], s["r1"], s["r2"])
        # ... and they are what the prover now holds
        assert from_limbs(cf.export(0, hip.IX_INSTANCE)) == Un and from_limbs(cf.export(1, hip.IX_INSTANCE)) == cn
        assert from_limbs(cf.export(0, hip.IX_FRESH_INSTANCE))[2:4] == [x0, x1]
    finally:
        cf.close()
