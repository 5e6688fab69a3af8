"""GPU parity for the folding path (SURVEY.md §8a rows W, V1, V2, M1, M2, F1', X1), through the C ABI:
witness generation vs the oracle's executor, sparse mat-vec vs the oracle, and whole folds checked
(a) by the library's own verify, (b) against the pinned IVC-state chain, (c) for the small `hash` circuit,
against an independent re-computation of every folded quantity with the oracle."""
import numpy as np
import pytest

from tests._oracle import from_limbs, r1cs_check, to_limbs, witness_execute
from tests.test_circuits import ORC_T, step_inputs
from vimz_amd import _lib
from vimz_amd.circuit import Circuit

pytestmark = pytest.mark.gpu

R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
OPS = ["hash", "grayscale", "contrast", "brightness", "blur", "sharpness", "resize", "redact"]


@pytest.fixture(scope="module")
def ctx():
    from vimz_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ck(ctx):
    b = ctx.bases_generate(_lib.CURVE_BN254_G1, 1 << 19)
    yield b
    b.free()


@pytest.fixture(scope="module")
def circuits():
    return {op: Circuit.for_resolution(op, "HD") for op in OPS}


@pytest.mark.parametrize("op", OPS)
def test_witness_matches_oracle_executor(ctx, ck, oracle, circuits, op):
    from vimz_amd import hip
    c = circuits[op]
    z0, inputs = step_inputs(op)
    P = hip.Prover(ctx, c, ck, max_batch=4)
    try:
        P.reset(z0)
        zw, zs, st = P.witness(np.stack(inputs[:4]))
        assert not st.any()
        z = list(z0)
        for i in range(4):
            status, want, z_out = witness_execute(oracle, c, z, inputs[i])
            assert status == 0
            assert from_limbs(zs[i]) == z and from_limbs(zs[i + 1]) == z_out
            diff = np.nonzero((zw[i] != want).any(axis=1))[0]
            assert diff.size == 0, f"{op} row {i}: {diff.size} wires differ, first {diff[:5]}"
            z = z_out
        # an unsatisfiable row is flagged, not silently accepted
        bad = np.stack(inputs[:2]).copy()
        if c.n_lane_groups:
            bad[1, -1, 0] ^= np.uint64(0x40)
            _, _, st = P.witness(bad, want_wires=False)
            assert st[0] == 0 and st[1] & 1
    finally:
        P.close()


@pytest.mark.parametrize("op", ["grayscale", "sharpness", "contrast"])
def test_spmv_matches_oracle(ctx, ck, oracle, circuits, op):
    """(A,B,C)·z on the shapes the bench runs (contrast = the headline workload), on a real witness and on dense random z."""
    from vimz_amd import hip
    c = circuits[op]
    z0, inputs = step_inputs(op)
    _, wires, _ = witness_execute(oracle, c, z0, inputs[0])
    rng = np.random.default_rng(3)
    dense = rng.integers(0, 1 << 62, size=wires.shape, dtype=np.uint64)
    dense[:, 3] &= np.uint64((1 << 58) - 1)
    P = hip.Prover(ctx, c, ck, max_batch=1)
    try:
        for z in (wires, dense):
            bad, want = r1cs_check(oracle, c, z, want_products=True)
            got = P.spmv(z)
            for g, w in zip(got, want):
                assert np.array_equal(g, w)
    finally:
        P.close()


@pytest.mark.parametrize("op", OPS)
def test_fold_ten_steps_verifies_and_tracks_state(ctx, ck, oracle, circuits, op):
    """DEMO_STEPS = 10 rows (vimz/src/lib.rs:9), as `make run-nova-snark-benchmarks` does in the reference."""
    from vimz_amd import hip
    c = circuits[op]
    z0, inputs = step_inputs(op)
    kw = dict(width=160) if op == "redact" else {}
    P = hip.Prover(ctx, c, ck, max_batch=4)          # 4+4+2: exercises batch boundaries
    try:
        P.reset(z0)
        P.fold(np.stack(inputs))
        assert P.verify() == 0
        inst = P.instance()
        assert inst["steps"] == 10
        z = list(z0)
        for i in range(10):
            ok, z = oracle.step_eval(ORC_T[op], z, inputs[i], **kw)
            assert ok
        assert from_limbs(inst["z"]) == z
    finally:
        P.close()


def test_fold_rejects_tampered_row(ctx, ck, circuits):
    from vimz_amd import hip
    c = circuits["contrast"]
    z0, inputs = step_inputs("contrast")
    bad = np.stack(inputs[:3]).copy()
    bad[2, -1, 0] ^= np.uint64(0x40)
    P = hip.Prover(ctx, c, ck, max_batch=2)
    try:
        P.reset(z0)
        with pytest.raises(_lib.VimzError) as e:
            P.fold(bad)
        assert e.value.code == _lib.ERR_UNSAT
    finally:
        P.close()


def _ro_point(pt):
    x, y = pt
    return [x & ((1 << 128) - 1), (x >> 128) | ((y & 1) << 126)]


def test_hash_circuit_fold_recomputed_by_oracle(ctx, ck, oracle, circuits):
    """Every quantity of a 3-step fold re-derived independently: commitments (oracle Pippenger over the downloaded
    key), cross term, Fiat–Shamir challenge (oracle Poseidon over the documented transcript), folded W, E, u."""
    from vimz_amd import hip
    c = circuits["hash"]
    z0, inputs = step_inputs("hash")
    n_aux = c.n_wires - 1 - 2 * c.len_z
    key = ck.download(0, max(n_aux, c.n_constraints))
    P = hip.Prover(ctx, c, ck, max_batch=2)
    try:
        P.reset(z0)
        P.fold(np.stack(inputs[:3]))
        inst = P.instance()
        z_run, E_run = P.running()
    finally:
        P.close()
    # --- oracle side
    zi = list(z0)
    wit, prods = [], []
    for i in range(3):
        st, w, zo = witness_execute(oracle, c, zi, inputs[i])
        bad, abc = r1cs_check(oracle, c, w, want_products=True)
        assert st == 0 and bad == -1
        wit.append(w); prods.append(abc); zi = zo
    aux0 = 1 + 2 * c.len_z
    commit = lambda v, n: oracle.msm(0, key[:n], v, threads=8)
    ro = oracle.poseidon([0x56494d7a, c.n_constraints, c.n_wires, c.len_z, 5, 128])
    zdig = 0
    for v in z0:
        zdig = oracle.poseidon([zdig, v])
    # step 0: running := fresh
    zchain = list(z0)
    st, _, z1 = witness_execute(oracle, c, zchain, inputs[0])
    for v in z1:
        zdig = oracle.poseidon([zdig, v])
    cW = commit(wit[0][aux0:], n_aux)
    ro = oracle.poseidon([ro] + _ro_point(cW) + [zdig])
    Z, (AZ, BZ, CZ) = wit[0], prods[0]
    E = np.zeros((c.n_constraints, 4), dtype=np.uint64)
    cE = (0, 0)
    u = 1
    znext = z1
    for i in (1, 2):
        st, _, znext2 = witness_execute(oracle, c, znext, inputs[i])
        for v in znext2:
            zdig = oracle.poseidon([zdig, v])
        znext = znext2
        a2, b2, c2 = prods[i]
        T = oracle.cross_term(0, AZ, BZ, CZ, u, a2, b2, c2, 1)
        cW2 = commit(wit[i][aux0:], n_aux)
        cT = commit(T, c.n_constraints)
        ro = oracle.poseidon([ro] + _ro_point(cW2) + _ro_point(cT) + [zdig])
        r = ro & ((1 << 128) - 1)
        Z = oracle.axpy(0, Z, r, wit[i]); E = oracle.axpy(0, E, r, T)
        AZ = oracle.axpy(0, AZ, r, a2); BZ = oracle.axpy(0, BZ, r, b2); CZ = oracle.axpy(0, CZ, r, c2)
        cW = oracle.curve_add(0, cW, oracle.curve_mul(0, cW2, r))
        cE = oracle.curve_add(0, cE, oracle.curve_mul(0, cT, r))
        u = (u + r) % R_MOD
    assert np.array_equal(z_run, Z)
    assert np.array_equal(E_run, E)
    assert from_limbs(inst["u"])[0] == u == from_limbs(z_run[0])[0]
    assert tuple(from_limbs(inst["comm_W"])) == cW
    assert tuple(from_limbs(inst["comm_E"])) == cE
    assert oracle.first_unsat(0, AZ, BZ, CZ, u=u, E=E) == -1


def test_segment_merge_matches_oracle_prover(ctx, ck, oracle, circuits):
    """Multi-GPU final fold on one GPU: two row segments folded by two provers, exported, merged (relaxed+relaxed NIFS);
    the merged accumulator verifies and equals the oracle-backed prover's bit for bit."""
    from tests._oracle_prover import OracleProver
    from vimz_amd import hip
    from vimz_amd.distributed import fold_sharded
    c = circuits["hash"]
    z0, inputs = step_inputs("hash")
    rows = np.stack(inputs[:5])
    A, B = hip.Prover(ctx, c, ck, max_batch=2), hip.Prover(ctx, c, ck, max_batch=2)
    try:
        zs = A.state_chain(z0, rows[:3])
        z_mid = from_limbs(zs[-1])
        A.reset(z0); A.fold(rows[:3])
        B.reset(z_mid); B.fold(rows[3:])
        assert A.verify() == 0 and B.verify() == 0
        blob_a, blob_b = A.export(), B.export()
        # segments must be adjacent and in row order: B after B, A after A, A after B are all refused and leave B untouched
        for bad in (blob_b, blob_a):
            with pytest.raises(_lib.VimzError):
                B.merge(bad)
        with pytest.raises(_lib.VimzError):
            A.merge(blob_a)
        assert B.instance()["steps"] == 2 and A.instance()["steps"] == 3
        A.merge(blob_b)
        assert A.verify() == 0
        inst = A.instance()
        z_run, E_run = A.running()
    finally:
        A.close(); B.close()
    n_aux = c.n_wires - 1 - 2 * c.len_z
    key = ck.download(0, max(n_aux, c.n_constraints))
    oa, ob = OracleProver(oracle, c, key), OracleProver(oracle, c, key)
    oa.reset(z0); oa.fold(rows[:3])
    assert oa.z == z_mid
    ob.reset(z_mid); ob.fold(rows[3:])
    oa.merge(ob.export())
    assert oa.verify() == 0
    assert inst["steps"] == 5 and from_limbs(inst["z"]) == oa.z
    assert from_limbs(inst["u"])[0] == oa.u
    assert tuple(from_limbs(inst["comm_W"])) == oa.cW and tuple(from_limbs(inst["comm_E"])) == oa.cE
    assert np.array_equal(z_run, oa.Z) and np.array_equal(E_run, oa.E)
    # and the single-process driver path
    P = hip.Prover(ctx, c, ck, max_batch=4)
    try:
        res = fold_sharded(P, rows, z0)
        assert res["verified"] and res["steps"] == 5
    finally:
        P.close()


def test_concurrent_local_segments_match_oracle(ck, oracle, circuits):
    """Three row segments folded concurrently on one GPU (own context + streams each, host threads) and merged on the
    device equal the oracle-backed segment folds merged in the same order."""
    from tests._oracle_prover import OracleProver
    from vimz_amd import hip
    from vimz_amd.distributed import fold_local_segments, segment_bounds
    c = circuits["hash"]
    z0, inputs = step_inputs("hash")
    rows = np.stack(inputs[:7])
    ctxs = [hip.Context(0) for _ in range(3)]
    provers = [hip.Prover(cx, c, ck, max_batch=2) for cx in ctxs]
    try:
        merged = fold_local_segments(provers, rows, z0)
        assert merged.verify() == 0
        inst = merged.instance()
        z_run, E_run = merged.running()
    finally:
        for p in provers:
            p.close()
        for cx in ctxs:
            cx.close()
    n_aux = c.n_wires - 1 - 2 * c.len_z
    key = ck.download(0, max(n_aux, c.n_constraints))
    acc, z = None, list(z0)
    for lo, hi in segment_bounds(7, 3):
        op = OracleProver(oracle, c, key)
        op.reset(z); op.fold(rows[lo:hi]); z = op.z
        if acc is None:
            acc = op
        else:
            acc.merge(op.export())
    assert inst["steps"] == 7 and from_limbs(inst["z"]) == acc.z and from_limbs(inst["u"])[0] == acc.u
    assert tuple(from_limbs(inst["comm_W"])) == acc.cW and tuple(from_limbs(inst["comm_E"])) == acc.cE
    assert np.array_equal(z_run, acc.Z) and np.array_equal(E_run, acc.E)


def test_resize_2to1_on_gpu(ctx, ck, oracle):
    """The 4K/8K resize geometry (2 rows -> 1) at a small width: GPU witness bit-equal to the oracle executor, fold verifies."""
    from tests.test_circuits import synthetic_resize_2to1
    from vimz_amd import hip
    c = Circuit("resize", 16, 8, 2, 1, 0)
    z0, inputs = synthetic_resize_2to1(steps=4)
    P = hip.Prover(ctx, c, ck, max_batch=3)
    try:
        P.reset(z0)
        zw, zs, st = P.witness(np.stack(inputs[:3]))
        assert not st.any()
        z = list(z0)
        for i in range(3):
            _, want, z = witness_execute(oracle, c, z, inputs[i])
            assert np.array_equal(zw[i], want)
        P.fold(np.stack(inputs))
        assert P.verify() == 0 and P.instance()["steps"] == 4
    finally:
        P.close()


@pytest.mark.parametrize("proof,image,op,kw", [("img2-contrast", "img2", "contrast", {"factor": 1.4}), ("img1-grayscale", "img1", "grayscale", {})])
def test_full_image_fold_ends_in_the_references_committed_state(ctx, ck, proof, image, op, kw):
    """All 720 rows of the reference's sample image folded on the GPU (two concurrent segments + on-device final fold):
    the accumulator verifies and the IVC state equals the final state inside the reference's committed proof
    (marketplace/proofs/*.proof, via tests/golden/kat.json)."""
    from tests import _data
    from vimz_amd import folding, hip, image_editor as ie
    from vimz_amd.distributed import fold_local_segments
    P = _data.kat()["proofs"][proof]
    inp = ie.build_input(op, _data.load_image(image), **kw)
    rows, z0 = folding.prepare_input(op, inp, "HD")
    assert len(rows) == P["steps"] == 720 and z0 == [int(v) for v in P["z0"]]
    c = Circuit.for_resolution(op, "HD")
    ctx2 = hip.Context(0)
    provers = [hip.Prover(ctx, c, ck, max_batch=32), hip.Prover(ctx2, c, ck, max_batch=32)]
    try:
        merged = fold_local_segments(provers, rows, z0)
        assert merged.verify() == 0
        inst = merged.instance()
        assert inst["steps"] == 720
        assert from_limbs(inst["z"]) == [int(v) for v in P["z_final"]]
    finally:
        for p in provers:
            p.close()
        ctx2.close()


def test_crop_steps_fold_from_supplied_witnesses(oracle):
    """BASELINE config #1 (crop_step HD, 672 k constraints) through the external-witness seam: the witnesses come from the
    oracle's executor; SpMV, both MSMs, folds and verify run on the GPU."""
    from tests import _data
    from tests._oracle import T_CROP
    from vimz_amd import hip, image_editor as ie
    c = Circuit.for_resolution("crop", "HD")
    fx = _data.rows10("crop")
    o = ie.hex_to_rows(fx["original"])
    z0 = [0, 0, fx["info"]]
    z, wits = list(z0), []
    for i in range(3):
        st, w, z = witness_execute(oracle, c, z, o[i])
        assert st == 0
        wits.append(w)
    cx = hip.Context(0)
    key = cx.bases_generate(_lib.CURVE_BN254_G1, 1 << 20)
    P = hip.Prover(cx, c, key, max_batch=2)
    try:
        P.reset(z0)
        P.fold_witness(np.stack(wits))
        assert P.verify() == 0
        inst = P.instance()
        assert inst["steps"] == 3 and from_limbs(inst["z"]) == z
        zz = list(z0)
        for i in range(3):
            ok, zz = oracle.step_eval(T_CROP, zz, o[i], width=128, width2=64, crop_h=480)
            assert ok
        assert zz == z
    finally:
        P.close(); key.free(); cx.close()


@pytest.mark.parametrize("y", [100, 0])
def test_crop_witness_and_folds_on_the_gpu(oracle, y):
    """crop_step on the GPU end to end: its witness needs an ahead-of-time pass (the hash of the cropped row enters the IVC
    state, and depends on step_in only through the predictable `info` counter).  With y = 100 the rows are outside the crop
    window (selector 0); with y = 0 they are inside it (selector 1: the cropped-row hash is absorbed).  (The literal circuit
    increments `info` by one per step, i.e. its x field — SURVEY F6 — so row_index stays 0.)"""
    from tests import _data
    from tests._oracle import T_CROP
    from vimz_amd import hip, image_editor as ie
    c = Circuit.for_resolution("crop", "HD")
    fx = _data.rows10("crop")
    o = ie.hex_to_rows(fx["original"])
    info = (int(fx["info"]) & 0xFFF) | (y << 12)          # x from the fixture, y as given, row_index 0
    z0 = [0, 0, info]
    cx = hip.Context(0)
    key = cx.bases_generate(_lib.CURVE_BN254_G1, 1 << 20)
    ck2 = cx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-secondary")
    P = hip.Prover(cx, c, key, max_batch=3)
    ivc = hip.IVC(cx, c, key, ck2, max_batch=3)
    try:
        P.reset(z0)
        zw, zs, st = P.witness(np.stack(o[:3]))
        assert not st.any()
        z = list(z0)
        for i in range(3):
            status, want, z_out = witness_execute(oracle, c, z, o[i])
            assert status == 0 and from_limbs(zs[i]) == z and from_limbs(zs[i + 1]) == z_out
            diff = np.nonzero((zw[i] != want).any(axis=1))[0]
            assert diff.size == 0, f"crop row {i}: {diff.size} wires differ, first {diff[:5]}"
            z = z_out
        zz = list(z0)
        for i in range(5):
            ok, zz = oracle.step_eval(T_CROP, zz, o[i], width=128, width2=64, crop_h=480)
            assert ok
        assert (zz[1] != 0) == (y == 0)                    # the cropped rows were (not) absorbed
        P.reset(z0); P.fold(np.stack(o[:5]))               # two batches: 3 + 2
        assert P.verify() == 0
        inst = P.instance()
        assert inst["steps"] == 5 and from_limbs(inst["z"]) == zz
        ivc.reset(z0); ivc.fold(np.stack(o[:5]))
        assert ivc.verify(5, z0) == 0 and ivc.state() == (zz, 5)
    finally:
        ivc.close(); P.close(); key.free(); ck2.free(); cx.close()


def test_fold_in_several_calls_equals_one_call(ctx, ck, circuits):
    """Ragged use of the ABI: folding 0, 3 and then 4 more rows gives exactly the accumulator of one 7-row call."""
    from vimz_amd import hip
    c = circuits["grayscale"]
    z0, inputs = step_inputs("grayscale")
    rows = np.stack(inputs[:7])
    A, B = hip.Prover(ctx, c, ck, max_batch=3), hip.Prover(ctx, c, ck, max_batch=5)
    try:
        A.reset(z0); A.fold(rows)
        B.reset(z0); B.fold(rows[:0]); B.fold(rows[:3]); B.fold(rows[3:])
        ia, ib = A.instance(), B.instance()
        for k in ("comm_W", "comm_E", "u", "z"):
            assert np.array_equal(ia[k], ib[k]), k
        assert ia["steps"] == ib["steps"] == 7 and A.verify() == 0 and B.verify() == 0
        za, ea = A.running(); zb, eb = B.running()
        assert np.array_equal(za, zb) and np.array_equal(ea, eb)
    finally:
        A.close(); B.close()


def test_largest_single_gpu_config_contrast_4k(oracle):
    """BASELINE config #3 (contrast at 4K, width 384: 914 593 constraints): three rows of a synthetic 4K image fold, verify,
    and track the oracle's step semantics."""
    from tests import _data
    from tests._oracle import T_CONTRAST
    from vimz_amd import folding, hip, image_editor as ie
    img = _data.load_image("img2")[:4]
    img = np.repeat(np.repeat(img, 3, axis=0), 3, axis=1)[:3]          # 3 rows x 3840 px
    inp = ie.build_input("contrast", img, factor=1.4)
    rows = np.stack([np.concatenate([inp["original"][i], inp["transformed"][i]]) for i in range(3)])
    c = Circuit.for_resolution("contrast", "4K")
    assert c.n_constraints == 914593 and rows.shape[1] == c.n_priv == 768
    cx = hip.Context(0)
    key = cx.bases_generate(_lib.CURVE_BN254_G1, 1 << 20)
    P = hip.Prover(cx, c, key, max_batch=2)
    try:
        z0 = [0, 0, 14]
        P.reset(z0)
        P.fold(rows)
        assert P.verify() == 0
        z = list(z0)
        for i in range(3):
            ok, z = oracle.step_eval(T_CONTRAST, z, rows[i], width=384)
            assert ok
        assert from_limbs(P.instance()["z"]) == z
    finally:
        P.close(); key.free(); cx.close()


def test_pixel_packing_on_the_device_matches_the_image_editor(ctx):
    """vimz_pack_pixels == our restatement of pyvimz's compress_by_rows / compress_by_blocks (itself pinned on fixtures minted by
    importing pyvimz, tests/test_oracle_golden.py) on the reference's sample images, RGB and grey, incl. a width that is not a
    multiple of ten."""
    from tests import _data
    from vimz_amd import image_editor as ie
    img = _data.load_image("img1")
    assert np.array_equal(ctx.pack_pixels(img), ie.compress_by_rows(img))
    grey = ie.convert_to_grayscale(img)
    assert np.array_equal(ctx.pack_pixels(grey), ie.compress_by_rows(grey))
    odd = img[:37, :1003]
    assert np.array_equal(ctx.pack_pixels(odd), ie.compress_by_rows(odd))
    assert np.array_equal(ctx.pack_pixels(img, block=40), ie.compress_by_blocks(img))
    assert ctx.pack_pixels(img).shape == (720, 128, 4)
