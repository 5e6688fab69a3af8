"""BASELINE.json's configurations at their full row width, in the benchmarked mode (Nova IVC), against the oracle
(VERDICT r1 "configs_untested"):
  #3  contrast_step on 4K (width 384, 914 592 constraints)
  #4  resize_step on 8K (768 -> 384 packed elements, 2 rows -> 1; vimz/src/transformation.rs:115-123)
  #5  proof set {contrast, brightness, sharpness, blur} on 4K, four IVC proofs folded side by side
For every configuration: the GPU witness of each row equals the oracle executor's wire for wire, the IVC proof is accepted by
vimz_ivc_verify for exactly (steps, z0), and the independent verifier built from the oracle (tests/test_gpu_ivc.py) accepts it."""
import numpy as np
import pytest

from tests import _data
from tests._oracle import from_limbs, witness_execute
from tests.test_gpu_ivc import oracle_verify
from vimz_amd import _lib
from vimz_amd.circuit import Circuit

pytestmark = pytest.mark.gpu

STEPS = 3


@pytest.fixture(scope="module")
def ctx():
    from vimz_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def keys(ctx):
    ck1 = ctx.bases_generate(_lib.CURVE_BN254_G1, 1 << 20)       # next_pow2 of the 4K circuits (0.75-0.98 M rows) + verifier circuit
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-secondary")
    yield ck1, ck2
    ck1.free(); ck2.free()


def _check_config(ctx, keys, oracle, op, res, check_commitments):
    from vimz_amd import hip
    ck1, ck2 = keys
    c = Circuit.for_resolution(op, res)
    rows, z0 = _data.config_rows(op, res, STEPS)
    assert rows.shape[1] == c.n_priv
    P = hip.Prover(ctx, c, ck1, max_batch=STEPS)
    try:
        P.reset(z0)
        zw, zs, st = P.witness(rows)
        assert not st.any(), f"{op} {res}: the GPU witness kernels flag a row as unsatisfiable"
        z = list(z0)
        for i in range(STEPS):
            status, want, z_next = witness_execute(oracle, c, z, rows[i])
            assert status == 0 and from_limbs(zs[i]) == z and from_limbs(zs[i + 1]) == z_next
            diff = np.nonzero((zw[i] != want).any(axis=1))[0]
            assert diff.size == 0, f"{op} {res} row {i}: {diff.size} wires differ from the oracle executor, first {diff[:5]}"
            z = z_next
    finally:
        P.close()
    ivc = hip.IVC(ctx, c, ck1, ck2, max_batch=2)
    try:
        ivc.reset(z0)
        ivc.fold(rows)                      # two batches: 2 + 1 rows
        assert ivc.verify(STEPS, z0) == 0
        assert ivc.state() == (z, STEPS)
        failed, z_exported = oracle_verify(oracle, ivc, ck1, ck2, STEPS, z0, check_commitments=check_commitments)
        assert failed == [] and z_exported == z
        info = ivc.info()
        assert info["step_constraints"] == c.n_constraints and info["step_wires"] == c.n_wires
    finally:
        ivc.close()
    return c


def test_config3_contrast_4k_ivc(ctx, keys, oracle):
    c = _check_config(ctx, keys, oracle, "contrast", "4K", check_commitments=True)     # commitments re-opened by the oracle's MSM (0.9 M points)
    assert c.n_constraints == 914_592 + 1                                               # SURVEY.md App. A projection (+1 linear)


def test_config4_resize_8k_ivc(ctx, keys, oracle):
    c = _check_config(ctx, keys, oracle, "resize", "8K", check_commitments=False)
    assert c.shape[:4] == (768, 384, 2, 1)
    assert c.n_constraints == 834_480


@pytest.mark.parametrize("op", ["brightness", "sharpness", "blur"])
def test_config5_members_4k_ivc(ctx, keys, oracle, op):
    _check_config(ctx, keys, oracle, op, "4K", check_commitments=False)


def test_config5_proof_set_4k_four_ivcs_side_by_side(keys, oracle):
    """The `--proof-set contrast,brightness,sharpness,blur` run of bench.py on one GPU: four different IVC proofs on four contexts
    folded concurrently from four host threads; each verifies for its own statement and ends in the oracle's state."""
    from vimz_amd import hip
    from vimz_amd.distributed import fold_concurrently
    ck1, ck2 = keys
    ops = ["contrast", "brightness", "sharpness", "blur"]
    ctxs = [hip.Context(0) for _ in ops]
    ivcs, jobs, want = [], [], []
    try:
        for cx, op in zip(ctxs, ops):
            c = Circuit.for_resolution(op, "4K")
            rows, z0 = _data.config_rows(op, "4K", STEPS)
            v = hip.IVC(cx, c, ck1, ck2, max_batch=2)
            v.reset(z0)
            ivcs.append(v); jobs.append((v, rows))
            z = list(z0)
            from tests.test_circuits import ORC_T
            for i in range(STEPS):
                ok, z = oracle.step_eval(ORC_T[op], z, rows[i], width=384)
                assert ok
            want.append((z0, z))
        fold_concurrently(jobs)
        for v, (z0, z) in zip(ivcs, want):
            assert v.verify(STEPS, z0) == 0 and v.state() == (z, STEPS)
    finally:
        for v in ivcs:
            v.close()
        for cx in ctxs:
            cx.close()


@pytest.mark.parametrize("op,res", [("resize", "8K"), ("contrast", "4K")])
def test_config_rows_sharded_into_three_segments_are_one_proof_object(keys, oracle, op, res):
    """BASELINE config 4's shape ("row-batches sharded ..., host-side final fold") on the one GPU of the box: six rows of the 8K
    resize circuit (and of the 4K contrast circuit) proven as THREE concurrent segments on three contexts and merged into ONE object
    (vimz_ivc_merge); the product's verifier and the oracle-side verifier (tests/_merge.py: replay of the records, relaxed relation of
    both folded instances over the 0.8-0.9 M-row shapes) accept it for (6 steps, z0), and it ends in the oracle's state."""
    from tests import _merge
    from tests.test_circuits import ORC_T
    from tests.test_gpu_ivc import _shape_digest
    from vimz_amd import hip
    from vimz_amd.distributed import fold_segments_merged
    ck1, ck2 = keys
    n = 6
    c = Circuit.for_resolution(op, res)
    rows, z0 = _data.config_rows(op, res, n)
    ctxs = [hip.Context(0) for _ in range(3)]
    ivcs = [hip.IVC(cx, c, ck1, ck2, max_batch=2) for cx in ctxs]
    m = None
    try:
        tm = {}
        m = fold_segments_merged(ivcs, rows, z0, tm)
        assert m.verify(n, z0) == 0 and m.verify(n - 1, z0) != 0
        z = list(z0)
        for i in range(n):
            ok, z = oracle.step_eval(ORC_T[op], z, rows[i], **({"width": 384} if res == "4K" else {"width": 768, "width2": 384, "rows_in": 2, "rows_out": 1}))
            assert ok
        zs, ze, steps = m.state()
        assert (zs, ze, steps) == ([int(x) for x in z0], z, n) and m.info()["segments"] == 3
        failed, acc = _merge.verify_merged(oracle, m, ivcs[0], ck1, ck2, n, z0, _shape_digest(ivcs[0], 0), _shape_digest(ivcs[0], 1), check_commitments=False)
        assert failed == [] and acc["ze"] == z
    finally:
        if m:
            m.close()
        for v in ivcs:
            v.close()
        for cx in ctxs:
            cx.close()
