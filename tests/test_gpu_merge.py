"""ONE proof object out of several row segments (vimz_ivc_merge*, SURVEY.md §8e / BASELINE.json north_star's "host-side sequential final
fold"; the reference's fold_input returns one RecursiveSNARK, vimz/src/nova_snark_backend/folding.rs:27-43): S IVC proofs of
contiguous row segments, folded concurrently, are merged by out-of-circuit NIFS on both curves.  The product's verifier must accept;
an independent verifier (tests/_merge.py: Python integers + hashlib + the oracle's Poseidon, curve arithmetic, R1CS check, MSM) must
accept; segments in the wrong order, foreign segments, other statements and tampered proofs must be rejected by both."""
import numpy as np
import pytest

from tests import _merge
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
    ck1 = ctx.bases_generate(_lib.CURVE_BN254_G1, 1 << 17)
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-secondary")
    yield ck1, ck2
    ck1.free(); ck2.free()


def _ints(limbs):
    return [int(a[0]) | int(a[1]) << 64 | int(a[2]) << 128 | int(a[3]) << 192 for a in limbs]


def _segments(ctx, c, ck1, ck2, steps, z0, cuts, ctxs=None):
    """IVC proofs of the row segments steps[cuts[j]:cuts[j+1]], each starting at the state the one before ends in."""
    from vimz_amd import hip
    ivcs, starts = [], []
    z = list(z0)
    for j in range(len(cuts) - 1):
        v = hip.IVC(ctxs[j] if ctxs else ctx, c, ck1, ck2, max_batch=4)
        v.reset(z)
        v.fold(steps[cuts[j]:cuts[j + 1]])
        assert v.verify(cuts[j + 1] - cuts[j], z) == 0
        starts.append(z)
        z = v.state()[0]
        ivcs.append(v)
    return ivcs, starts


@pytest.mark.parametrize("op,cuts", [("hash", [0, 3, 7, 10]), ("grayscale", [0, 1, 6, 10]), ("hash", [0, 4, 5, 6, 10])])
def test_merged_proof_verifies_and_the_oracle_verifier_accepts(ctx, keys, oracle, op, cuts):
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution(op, "HD")
    z0, inputs = step_inputs(op)
    steps = np.stack(inputs)
    ivcs, starts = _segments(ctx, c, ck1, ck2, steps, z0, cuts)
    whole = hip.IVC(ctx, c, ck1, ck2, max_batch=4)
    m = None
    try:
        whole.reset(z0); whole.fold(steps)
        m = hip.MergedProof.of(ivcs)
        assert m.verify(10, z0) == 0
        zs, ze, n = m.state()
        assert (zs, n) == ([int(x) for x in z0], 10) and ze == whole.state()[0]       # one statement: the whole image's
        info = m.info()
        assert info["segments"] == len(cuts) - 1 and info["ops"] == 2 * (len(cuts) - 1) - 1
        # another statement is not what was proven
        assert m.verify(9, z0) & 4096 and m.verify(10, [x + 1 for x in z0]) & 4096
        # the independent verifier
        d1, d2 = _shape_digest(ivcs[0], 0), _shape_digest(ivcs[0], 1)
        failed, acc = _merge.verify_merged(oracle, m, ivcs[0], ck1, ck2, 10, z0, d1, d2, check_commitments=(op == "hash"))
        assert failed == [] and acc["ze"] == ze
        assert "step count" in _merge.verify_merged(oracle, m, ivcs[0], ck1, ck2, 11, z0, d1, d2, check_commitments=False)[0]
        # the product's host arithmetic agrees with the oracle's on the folded instances
        for side, key in ((0, "P"), (1, "Q")):
            inst = _ints(m.export(side, hip.IX_INSTANCE))
            cW, cE, u, X0, X1 = acc[key]
            assert inst == [cW[0], cW[1], cE[0], cE[1], u, X0, X1]
        # the segments are left as they were: they still verify and can fold on
        assert [v.verify(cuts[j + 1] - cuts[j], starts[j]) for j, v in enumerate(ivcs)] == [0] * len(ivcs)
    finally:
        if m:
            m.close()
        for v in ivcs:
            v.close()
        whole.close()


def test_wrong_order_foreign_and_overlapping_segments_are_refused(ctx, keys):
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("hash", "HD")
    z0, inputs = step_inputs("hash")
    steps = np.stack(inputs)
    ivcs, _ = _segments(ctx, c, ck1, ck2, steps, z0, [0, 3, 6, 10])
    foreign = hip.IVC(ctx, c, ck1, ck2, max_batch=4)           # rows 6.. proven from ANOTHER start state
    m = None
    try:
        foreign.reset([z0[0] + 5]); foreign.fold(steps[6:])
        m = hip.MergedProof(ivcs[0])
        for bad in (ivcs[2], ivcs[0], foreign):               # skips a segment / repeats one / not of this chain
            with pytest.raises(_lib.VimzError):
                m.merge(bad)
        assert m.info()["segments"] == 1 and m.verify(3, z0) == 0      # a refused merge leaves the object as it was
        m.merge(ivcs[1])
        with pytest.raises(_lib.VimzError):
            m.merge(foreign)
        m.merge(ivcs[2])
        assert m.verify(10, z0) == 0
        # an IVC that has folded nothing cannot be merged or start a merged proof
        empty = hip.IVC(ctx, c, ck1, ck2, max_batch=4)
        try:
            empty.reset(m.state()[1])
            with pytest.raises(_lib.VimzError):
                m.merge(empty)
            with pytest.raises(_lib.VimzError):
                hip.MergedProof(empty)
        finally:
            empty.close()
    finally:
        if m:
            m.close()
        for v in ivcs:
            v.close()
        foreign.close()


def test_saved_proof_loads_elsewhere_merges_on_and_tampering_is_caught(ctx, keys, oracle):
    """Two merged proofs of adjacent runs of segments (what two GPUs would each produce) travel as bytes, are loaded against a
    verifier key built independently, merged into one and verified; flipped bits anywhere in the blob are rejected at load or fail
    verification; a record swapped for another segment's fails the hash / adjacency checks."""
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("hash", "HD")
    z0, inputs = step_inputs("hash")
    steps = np.stack(inputs)
    ctx2 = hip.Context(0)
    ivcs, _ = _segments(ctx, c, ck1, ck2, steps, z0, [0, 2, 5, 8, 10], ctxs=[ctx, ctx2, ctx, ctx2])
    vk = hip.IVC(ctx, c, ck1, ck2, max_batch=1)
    objs = []
    try:
        a = hip.MergedProof.of(ivcs[:2]); objs.append(a)
        b = hip.MergedProof.of(ivcs[2:]); objs.append(b)
        blob_a, blob_b = a.save(), b.save()
        la = hip.MergedProof.load(vk, blob_a); objs.append(la)
        lb = hip.MergedProof.load(vk, blob_b); objs.append(lb)
        assert la.verify(5, z0) == 0 and lb.verify(5, a.state()[1]) == 0
        la.merge(lb)
        assert la.verify(10, z0) == 0 and la.info()["segments"] == 4
        d1, d2 = _shape_digest(vk, 0), _shape_digest(vk, 1)
        failed, acc = _merge.verify_merged(oracle, la, vk, ck1, ck2, 10, z0, d1, d2)
        assert failed == []
        with pytest.raises(_lib.VimzError):
            lb.merge(la)                                        # the other way round the runs are not adjacent
        # the whole thing once more through bytes
        again = hip.MergedProof.load(vk, la.save()); objs.append(again)
        assert again.verify(10, z0) == 0
        assert (again.records() == la.records()).all()
        # tampering
        blob = la.save()
        rec_bytes = 8 * len(la.records())
        for where in (8 * 9 + 3, 8 * 40, rec_bytes - 20, rec_bytes + 64, rec_bytes + (len(blob) - rec_bytes) // 2, len(blob) - 40):
            bad = blob.copy()
            bad[where] ^= 1
            try:
                t = hip.MergedProof.load(vk, bad)
            except _lib.VimzError:
                continue                                        # (off-curve point, unreduced element, malformed header)
            try:
                assert t.verify(10, z0) != 0, where
            finally:
                t.close()
        with pytest.raises(_lib.VimzError):
            hip.MergedProof.load(vk, blob[:len(blob) - 8])
        # the oracle verifier is not vacuous either: a record of another segment in place of the second one
        words = la.records().copy()
        rec = _merge.parse_records(words, c.len_z)
        rec["segs"][1], rec["segs"][2] = rec["segs"][2], rec["segs"][1]
        f2, _ = _merge.replay(oracle, rec, d1, d2, c.len_z)
        assert "segments not adjacent" in f2
    finally:
        for o in objs:
            o.close()
        for v in ivcs:
            v.close()
        vk.close()
        ctx2.close()


def test_compressed_merged_proof(ctx, oracle):
    """vimz_ivc_merged_compress: one Spartan + IPA argument for the folded primary and one for the folded secondary instance; the
    verifier replays the records and accepts only the statement that was proven."""
    from vimz_amd import hip
    ck1 = ctx.bases_generate(_lib.CURVE_BN254_G1, 1 << 15)
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-secondary")
    c = Circuit.for_resolution("hash", "HD")
    z0, inputs = step_inputs("hash")
    steps = np.stack(inputs)
    ivcs, _ = _segments(ctx, c, ck1, ck2, steps, z0, [0, 4, 7, 10])
    vk = hip.IVC(ctx, c, ck1, ck2, max_batch=1)
    m = None
    try:
        m = hip.MergedProof.of(ivcs)
        proof, t = m.compress()
        assert len(proof) < 64 * 1024
        assert hip.MergedProof.verify_compressed(vk, proof, 10, z0) == 0
        assert hip.MergedProof.verify_compressed(vk, proof, 9, z0) & 4096
        assert hip.MergedProof.verify_compressed(vk, proof, 10, [z0[0] + 1]) & 4096
        words = len(proof) // 8
        for where in (8 * 30, 8 * 60, 8 * (words // 2), 8 * (2 * words // 3), len(proof) - 24):
            bad = proof.copy()
            bad[where] ^= 1
            assert hip.MergedProof.verify_compressed(vk, bad, 10, z0) != 0, where
    finally:
        if m:
            m.close()
        for v in ivcs:
            v.close()
        vk.close()
        ck1.free(); ck2.free()


def test_full_image_as_three_concurrent_segments_is_one_proof_ending_in_the_references_state(ctx):
    """All 720 rows of the reference's sample image (img2, contrast 1.4) as three row segments folded concurrently on one GPU and
    merged: ONE object that verifies for (720 steps, the transformation's z0) and whose final state is the one inside the
    reference's committed proof (marketplace/proofs/img2-contrast.proof via tests/golden/kat.json)."""
    from tests import _data
    from vimz_amd import folding, hip, image_editor as ie
    from vimz_amd.distributed import fold_concurrently, ivc_segments
    P = _data.kat()["proofs"]["img2-contrast"]
    inp = ie.build_input("contrast", _data.load_image("img2"), factor=1.4)
    rows, z0 = folding.prepare_input("contrast", inp, "HD")
    circuit, params = folding.prepare_folding(ctx, "contrast", "HD")
    ck2 = params.secondary_key()
    cx = [ctx, hip.Context(0), hip.Context(0)]
    ivcs = [hip.IVC(c_, circuit, params.ck, ck2, max_batch=32) for c_ in cx]
    m = None
    try:
        segs = ivc_segments(ivcs, rows, z0)
        for v, r, z in segs:
            v.reset(z)
        fold_concurrently([(v, r) for v, r, z in segs])
        m = hip.MergedProof.of(ivcs)
        assert m.verify(720, z0) == 0
        zs, ze, n = m.state()
        assert n == 720 and ze == [int(x) for x in P["z_final"]]
        assert m.verify(719, z0) != 0
    finally:
        if m:
            m.close()
        for v in ivcs:
            v.close()
        params.free()
        for c_ in cx[1:]:
            c_.close()


@pytest.mark.parametrize("op,n", [("contrast", 10), ("blur", 10), ("resize", 10), ("hash", 10)])
def test_state_chain_in_two_parts_equals_the_one_call_chain(ctx, op, n):
    """vimz_ivc_row_digests (any rank can hash any rows) + vimz_ivc_chain_from_digests (the serial part, host only) give the states
    vimz_ivc_state_chain gives, for a run of rows cut in two at any point — what lets the ranks of a sharded proof hash their own
    rows side by side (prove_sharded); the GPU hash pass and the host pool path (short inputs) both; crop reports stride 0."""
    from vimz_amd import folding, hip
    c, params = folding.prepare_folding(ctx, op, "HD")
    ck1, ck2 = params.ck, params.secondary_key()
    z0, inputs = step_inputs(op)
    rows = np.stack(inputs)[:n]
    a, b = hip.IVC(ctx, c, ck1, ck2, max_batch=4), hip.IVC(ctx, c, ck1, ck2, max_batch=4)
    try:
        want = a.state_chain(z0, rows)
        assert a.digest_stride() > 0
        d1, d2 = a.row_digests(rows[:4]), b.row_digests(rows[4:])            # "two ranks"
        z_mid = a.chain_from_digests(z0, rows[:4], d1)
        assert (z_mid == want[:5]).all()
        z_end = b.chain_from_digests(_ints(z_mid[-1]), rows[4:], d2)
        assert (z_end == want[4:]).all()
        big = np.concatenate([rows] * 12)[:100]                               # > 96 rows: the GPU hash pass
        wb = a.state_chain(z0, big)
        assert (a.chain_from_digests(z0, big, a.row_digests(big)) == wb).all()
        bad = d1.copy(); bad[0, 0] = 0xFFFFFFFFFFFFFFFF                       # an unreduced digest element is refused
        with pytest.raises(_lib.VimzError):
            a.chain_from_digests(z0, rows[:4], bad)
    finally:
        a.close(); b.close(); params.free()


def test_crop_has_no_state_independent_digests(ctx):
    from vimz_amd import folding, hip
    circuit, params = folding.prepare_folding(ctx, "crop", "HD")
    v = hip.IVC(ctx, circuit, params.ck, params.secondary_key(), max_batch=2)
    try:
        assert v.digest_stride() == 0
        with pytest.raises(_lib.VimzError):
            v.row_digests(np.zeros((1, circuit.n_priv, 4), dtype=np.uint64))
    finally:
        v.close(); params.free()


def test_fold_segments_in_one_call_edge_cases(ctx, keys, oracle):
    """vimz_ivc_fold_segments: fewer rows than segments (the spare IVCs are left out), the same object as the hand-made sequence
    (state chain, reset, fold, create, merge) gives, two segments on one context refused, an unsatisfiable row reported as such and
    no object returned."""
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("grayscale", "HD")
    z0, inputs = step_inputs("grayscale")
    rows = np.stack(inputs)
    cx = [ctx, hip.Context(0), hip.Context(0)]
    ivcs = [hip.IVC(c_, c, ck1, ck2, max_batch=4) for c_ in cx]
    objs = []
    try:
        m2, t = hip.MergedProof.fold_segments(ivcs, rows[:2], z0); objs.append(m2)
        assert m2.info()["segments"] == 2 and m2.verify(2, z0) == 0 and set(t) == {"state_chain_s", "merge_s", "total_s"}
        m, _ = hip.MergedProof.fold_segments(ivcs, rows, z0); objs.append(m)
        assert m.verify(10, z0) == 0 and m.info()["segments"] == 3
        # the same by hand
        hand = [hip.IVC(c_, c, ck1, ck2, max_batch=4) for c_ in cx]
        z, cuts = list(z0), [0, 4, 7, 10]
        for k, v in enumerate(hand):
            v.reset(z); v.fold(rows[cuts[k]:cuts[k + 1]]); z = v.state()[0]
        mh = hip.MergedProof.of(hand); objs.append(mh)
        assert (mh.records() == m.records()).all()                     # same segments, same commitments, same challenges
        for v in hand:
            v.close()
        # its verifier key (hand[0]) is gone: the object is orphaned — calls on it fail cleanly, closing it is safe
        with pytest.raises(_lib.VimzError):
            mh.info()
        assert mh.verify(10, z0) == 8192
        with pytest.raises(_lib.VimzError):
            hip.MergedProof.fold_segments([ivcs[0], ivcs[0]], rows, z0)
        bad = rows.copy(); bad[8, 200, 0] ^= np.uint64(0xFF)
        with pytest.raises(_lib.VimzError) as e:
            hip.MergedProof.fold_segments(ivcs, bad, z0)
        assert e.value.code == _lib.ERR_UNSAT
        m3, _ = hip.MergedProof.fold_segments(ivcs, rows, z0); objs.append(m3)       # the provers are usable afterwards
        assert m3.verify(10, z0) == 0
        # short segments have their head rows hashed once, ahead of their folds (merge.hip: head_precompute); the other schedule — row
        # digests and head batches apart — gives the same object
        import os
        os.environ["VIMZ_DEBUG_NO_HEAD_PRECOMPUTE"] = "1"
        try:
            m4, _ = hip.MergedProof.fold_segments(ivcs, rows, z0); objs.append(m4)
        finally:
            del os.environ["VIMZ_DEBUG_NO_HEAD_PRECOMPUTE"]
        assert m4.verify(10, z0) == 0 and (m4.records() == m.records()).all() and (m3.records() == m.records()).all()
    finally:
        for o in objs:
            o.close()
        for v in ivcs:
            v.close()
        for c_ in cx[1:]:
            c_.close()


def test_a_run_of_rows_begun_before_its_start_state_is_known(ctx, keys, oracle):
    """vimz_ivc_fold_segments_begin / _pending_digests / _pending_start / _pending_finish — a rank of a sharded proof: the segments' folds begin, the rows'
    digests come out of the folds' own chain passes (equal to vimz_ivc_row_digests' on the chains' slots), the start state arrives later; the object is the
    one vimz_ivc_fold_segments makes from that state (same records), also for a run that does not start at z_0; every schedule with and without the library's
    host-evaluated head batch gives the same object; a run cancelled before its state arrived leaves the provers usable; an unsatisfiable row is reported."""
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution("grayscale", "HD")
    z0, inputs = step_inputs("grayscale")
    rows = np.stack(inputs)
    cx = [ctx, hip.Context(0), hip.Context(0)]
    ivcs = [hip.IVC(c_, c, ck1, ck2, max_batch=4) for c_ in cx]
    objs = []
    try:
        ref, _ = hip.MergedProof.fold_segments(ivcs, rows, z0); objs.append(ref)
        assert ref.verify(10, z0) == 0
        want = np.asarray(ivcs[0].row_digests(rows))      # (before the run begins: a prover's context is taken while its fold waits for the start state)
        p = hip.MergedProof.fold_segments_begin(ivcs, rows)
        dg = p.digests()
        want = want.reshape(dg.shape)
        nz = np.any(want.reshape(len(rows), -1, 4) != 0, axis=2)                 # the slots row_digests fills: the chains' outputs
        assert (dg[nz] == want[nz]).all() and not np.any(dg[~nz])
        p.start(z0)
        m, t = p.finish(); objs.append(m)
        assert m.verify(10, z0) == 0 and m.verify(9, z0) != 0 and (m.records() == ref.records()).all()
        # a run in the middle of an image: rows 3..9 from the state after rows 0..2 (what rank 1 of two ranks does)
        zs = ivcs[0].chain_from_digests(z0, rows[:3], want[:3])
        z3 = [sum(int(zs[-1][i][k]) << (64 * k) for k in range(4)) for i in range(len(z0))]
        p = hip.MergedProof.fold_segments_begin(ivcs, rows[3:])
        assert (p.digests()[nz[3:]] == want[3:][nz[3:]]).all()
        p.start(z3)
        m2, _ = p.finish(); objs.append(m2)
        ref2, _ = hip.MergedProof.fold_segments(ivcs, rows[3:], z3); objs.append(ref2)
        assert m2.verify(7, z3) == 0 and m2.verify(7, z0) != 0 and (m2.records() == ref2.records()).all()
        # cancelled before the state arrived: nothing is returned, the provers fold again afterwards
        p = hip.MergedProof.fold_segments_begin(ivcs, rows)
        p.digests()
        p.cancel()
        m3, _ = hip.MergedProof.fold_segments(ivcs, rows, z0); objs.append(m3)
        assert (m3.records() == ref.records()).all()
        # an unsatisfiable row: the error comes back from finish, no object
        bad = rows.copy(); bad[8, 200, 0] ^= np.uint64(0xFF)
        p = hip.MergedProof.fold_segments_begin(ivcs, bad)
        p.start(z0)
        with pytest.raises(_lib.VimzError) as e:
            p.finish()
        assert e.value.code == _lib.ERR_UNSAT
        m4, _ = hip.MergedProof.fold_segments(ivcs, rows, z0); objs.append(m4)
        assert m4.verify(10, z0) == 0
        assert hip.head_rows_policy(7) in (0, 7) and hip.head_rows_policy(1000) == 0 and hip.head_rows_policy(100, segments=True) == 0
    finally:
        for o in objs:
            o.close()
        for v in ivcs:
            v.close()
        for c_ in cx[1:]:
            c_.close()
