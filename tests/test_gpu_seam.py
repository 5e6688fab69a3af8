"""The standalone SpMV / commit_T seam (include/vimz_hip.h: vimz_r1cs_upload, vimz_spmv3, vimz_commit_T):
a caller-supplied shape as COO triplets — what nova-snark's R1CSShape holds — against the oracle's spmv / cross_term / msm."""
import numpy as np
import pytest

from tests._oracle import from_limbs, r1cs_check, to_limbs, witness_execute
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


def _coo_from_circuit(c, rng):
    """The step circuit's matrices as shuffled (row, col, value) triplets with explicit canonical coefficients."""
    dict_canon = c.export("DICT_CANON", np.uint64).reshape(-1, 4)
    out = []
    for m in "ABC":
        rp, col, coef = c.csr(m)
        rows = np.repeat(np.arange(len(rp) - 1, dtype=np.uint32), np.diff(rp).astype(np.int64))
        vals = dict_canon[coef]
        perm = rng.permutation(len(rows))
        out.append((rows[perm], col[perm], vals[perm]))
    return out


@pytest.mark.parametrize("op", ["hash", "grayscale"])
def test_multiply_vec_and_commit_T_match_the_oracle(ctx, oracle, op):
    from vimz_amd import hip
    c = Circuit.for_resolution(op, "HD")
    rng = np.random.default_rng(5)
    A, B, Cm = _coo_from_circuit(c, rng)
    S = hip.R1CSShape(ctx, _lib.FIELD_BN254_FR, c.n_constraints, c.n_wires, A, B, Cm)
    ck = ctx.bases_generate(_lib.CURVE_BN254_G1, c.n_constraints)
    try:
        z0, inputs = step_inputs(op)
        _, w_fresh, _ = witness_execute(oracle, c, z0, inputs[0])
        dense = rng.integers(0, 1 << 62, size=w_fresh.shape, dtype=np.uint64)
        dense[:, 3] &= np.uint64((1 << 58) - 1)
        prods = {}
        for name, z in (("fresh", w_fresh), ("running", dense)):
            zd = ctx.vec_from_host(_lib.FIELD_BN254_FR, z)
            got = S.multiply_vec(zd)
            _, want = r1cs_check(oracle, c, z, want_products=True)
            for g, w in zip(got, want):
                assert np.array_equal(g.download(), w), f"{op}: (A,B,C)·z differs on the {name} vector"
                g.free()
            prods[name] = (zd, want)
        u1 = int(rng.integers(1, 1 << 62)) << 64 | 12345          # a running instance's relaxation scalar
        (z1, (a1, b1, c1)), (z2, (a2, b2, c2)) = prods["running"], prods["fresh"]
        T, comm = S.commit_T(ck, z1, u1, z2, 1)
        want_T = oracle.cross_term(0, a1, b1, c1, u1, a2, b2, c2, 1)
        assert np.array_equal(T.download(), want_T)
        bases = ck.download(0, c.n_constraints)
        assert tuple(from_limbs(comm)) == oracle.msm(0, bases, want_T, threads=8)
        T.free(); z1.free(); z2.free()
    finally:
        S.free(); ck.free()


def _random_coo(rng, nrows, ncols, modulus, long_rows=()):
    """Random sparse matrix as shuffled triplets (no duplicate positions): 1-4 terms per row, a few long rows, coefficients that are
    small, negative-small or full-width."""
    rows, cols, vals = [], [], []
    for r in range(nrows):
        k = 120 if r in long_rows else int(rng.integers(1, 5))
        for c in rng.choice(ncols, size=k, replace=False):
            kind = int(rng.integers(0, 4))
            v = 1 if kind == 0 else int(rng.integers(2, 1 << 16)) if kind == 1 else modulus - int(rng.integers(1, 1 << 16)) if kind == 2 else int.from_bytes(rng.bytes(32), "little") % modulus
            rows.append(r); cols.append(int(c)); vals.append(v)
    perm = rng.permutation(len(rows))
    return np.asarray(rows, dtype=np.uint32)[perm], np.asarray(cols, dtype=np.uint32)[perm], to_limbs([vals[i] for i in perm])


def _csr(rows, cols, vals, nrows):
    order = np.argsort(rows, kind="stable")
    rp = np.zeros(nrows + 1, dtype=np.uint32)
    np.add.at(rp, rows.astype(np.int64) + 1, 1)
    return np.cumsum(rp).astype(np.uint32), cols[order], vals[order]


@pytest.mark.parametrize("curve", [_lib.CURVE_PALLAS, _lib.CURVE_VESTA])
def test_seam_over_the_pasta_fields(ctx, oracle, curve):
    """north_star names the Pasta cycle: (A,B,C)·z and commit_T for a caller-supplied shape over Pallas' / Vesta's scalar field, with the
    commitment on that curve, against the oracle's spmv / cross_term / msm in the same field."""
    from vimz_amd import hip
    fid = _lib.CURVE_SCALAR_FIELD[curve]
    q = oracle.modulus[fid]
    rng = np.random.default_rng(40 + curve)
    nrows, ncols = 6000, 5000
    mats = [_random_coo(rng, nrows, ncols, q, long_rows=(7, 4100)) for _ in range(3)]
    S = hip.R1CSShape(ctx, fid, nrows, ncols, *mats)
    ck = ctx.bases_upload(curve, oracle.seq_bases(curve, nrows))
    try:
        zs = []
        for _ in range(2):
            z = to_limbs([int.from_bytes(rng.bytes(32), "little") % q for _ in range(ncols)])
            zs.append((ctx.vec_from_host(fid, z), z))
        prods = []
        for zd, z in zs:
            got = S.multiply_vec(zd)
            want = [oracle.spmv(fid, nrows, ncols, *_csr(*m, nrows), z) for m in mats]
            for g, w in zip(got, want):
                assert np.array_equal(g.download(), w)
                g.free()
            prods.append(want)
        u1 = int.from_bytes(rng.bytes(31), "little")
        T, comm = S.commit_T(ck, zs[0][0], u1, zs[1][0], 1)
        want_T = oracle.cross_term(fid, *prods[0], u1, *prods[1], 1)
        assert np.array_equal(T.download(), want_T)
        assert tuple(from_limbs(comm)) == oracle.msm(curve, oracle.seq_bases(curve, nrows), want_T, threads=8)
        T.free()
        for zd, _ in zs:
            zd.free()
    finally:
        S.free(); ck.free()


def _toy_circuit(rng, q, n_in, n_mul, n_lin):
    """A satisfiable R1CS over GF(q) in the layout z = [W | u | x0, x1]: input wires, products of earlier wires (one constraint each),
    linear combinations (lin · u = w, so they stay consistent in relaxed form), and two public outputs.  Returns the three matrices as
    triplet lists and a witness generator."""
    m = n_in + n_mul + n_lin                       # witness wires; u at column m, X at m+1, m+2
    A, B, Cm, ops = [], [], [], []
    row = 0
    for k in range(n_in, m):
        a, b = int(rng.integers(0, k)), int(rng.integers(0, k))
        ca, cb = int(rng.integers(1, 1 << 20)), q - int(rng.integers(1, 1 << 20))
        if k < n_in + n_mul:
            A.append((row, a, ca)); B.append((row, b, cb)); Cm.append((row, k, 1)); ops.append(("mul", k, a, b, ca, cb))
        else:
            A.append((row, a, ca)); A.append((row, b, cb) if b != a else (row, (a + 1) % k, cb)); B.append((row, m, 1)); Cm.append((row, k, 1))
            ops.append(("lin", k, a, b if b != a else (a + 1) % k, ca, cb))
        row += 1
    for j, src in enumerate((m - 1, m - 2)):       # public outputs: w_src · u = x_j
        A.append((row, src, 1)); B.append((row, m, 1)); Cm.append((row, m + 1 + j, 1)); row += 1

    def witness():
        w = [int.from_bytes(rng.bytes(32), "little") % q for _ in range(n_in)] + [0] * (m - n_in)
        for kind, k, a, b, ca, cb in ops:
            w[k] = (ca * w[a] % q) * (cb * w[b] % q) % q if kind == "mul" else (ca * w[a] + cb * w[b]) % q
        return w, [w[m - 1], w[m - 2]]

    def coo(T):
        return (np.asarray([t[0] for t in T], dtype=np.uint32), np.asarray([t[1] for t in T], dtype=np.uint32), to_limbs([t[2] for t in T]))
    return row, m + 3, m, coo(A), coo(B), coo(Cm), witness


@pytest.mark.parametrize("curve", [_lib.CURVE_PALLAS, _lib.CURVE_VESTA])
def test_nifs_over_the_pasta_cycle_through_the_seam(ctx, oracle, curve):
    """Nova's non-interactive folding (NIFS::prove as RecursiveSNARK::prove_step runs it, folding.rs:35-41) over Pallas / Vesta with
    nothing but the C-ABI seam: commitments (vimz_msm_vec), cross term and its commitment (vimz_commit_T), the folds of W and E
    (vimz_vec_axpy); the folded commitments and scalars on the host.  Five fresh instances of a satisfiable toy circuit are folded;
    the oracle then checks that the running instance satisfies the relaxed relation and that its two commitments open to W and E."""
    from vimz_amd import hip
    fid = _lib.CURVE_SCALAR_FIELD[curve]
    q = oracle.modulus[fid]
    rng = np.random.default_rng(1000 + curve)
    nrows, ncols, m, A, B, Cm, witness = _toy_circuit(rng, q, n_in=64, n_mul=3000, n_lin=500)
    S = hip.R1CSShape(ctx, fid, nrows, ncols, A, B, Cm)
    nk = max(nrows, m)
    bases = oracle.seq_bases(curve, nk)
    ck = ctx.bases_upload(curve, bases)
    zrun = ctx.vec_alloc(fid, ncols)              # running instance: z = [W | u | X], all zero to begin with
    E = ctx.vec_alloc(fid, nrows)
    u_run, X_run, cW_run, cE_run = 0, [0, 0], (0, 0), (0, 0)
    try:
        for step in range(5):
            w, X = witness()
            z2 = ctx.vec_from_host(fid, to_limbs(w + [1] + X))
            cW2 = tuple(from_limbs(ctx.msm_vec(ck, z2, n=m)))
            T, cT = S.commit_T(ck, zrun, u_run, z2, 1)
            cT = tuple(from_limbs(cT))
            r = (int.from_bytes(rng.bytes(16), "little") | 1 << 127) % q       # any challenge: the transcript is the caller's business
            hip.vec_axpy(ctx, zrun, r, z2)                                     # W, u and X in one pass (z holds all three)
            hip.vec_axpy(ctx, E, r, T)
            u_run = (u_run + r) % q
            X_run = [(a + r * b) % q for a, b in zip(X_run, X)]
            cW_run = oracle.curve_add(curve, cW_run, oracle.curve_mul(curve, cW2, r))
            cE_run = oracle.curve_add(curve, cE_run, oracle.curve_mul(curve, cT, r))
            T.free(); z2.free()
        zh, Eh = zrun.download(), E.download()
        assert from_limbs(zh[m]) == [u_run] and from_limbs(zh[m + 1:]) == X_run
        prods = [oracle.spmv(fid, nrows, ncols, *_csr(*M, nrows), zh) for M in (A, B, Cm)]
        assert oracle.first_unsat(fid, *prods, u=u_run, E=Eh) == -1           # Az∘Bz = u·Cz + E
        assert oracle.msm(curve, bases[:m], zh[:m], threads=8) == cW_run
        assert oracle.msm(curve, bases[:nrows], Eh, threads=8) == cE_run
        bad = Eh.copy(); bad[5, 0] ^= np.uint64(1)
        assert oracle.first_unsat(fid, *prods, u=u_run, E=bad) == 5
    finally:
        S.free(); ck.free(); zrun.free(); E.free()


@pytest.mark.parametrize("curve", [_lib.CURVE_PALLAS, _lib.CURVE_VESTA, _lib.CURVE_BN254_G1])
def test_relaxed_accumulator_over_the_seam(ctx, oracle, curve):
    """vimz_amd.nifs.RelaxedAccumulator (the product's NIFS over a caller-supplied shape, device calls only) against the oracle: after
    six folds its own verify() passes, the oracle agrees on the relaxed relation and on both openings and re-derives the folded
    commitments from the per-step values with its own curve arithmetic; a tampered error vector or witness is rejected."""
    from vimz_amd import hip, nifs
    fid = _lib.CURVE_SCALAR_FIELD[curve]
    q = oracle.modulus[fid]
    rng = np.random.default_rng(2000 + curve)
    nrows, ncols, m, A, B, Cm, witness = _toy_circuit(rng, q, n_in=32, n_mul=2500, n_lin=300)
    acc = nifs.RelaxedAccumulator(ctx, curve, nrows, m, 2, A, B, Cm, q)
    try:
        cW, cE = (0, 0), (0, 0)
        for _ in range(6):
            w, X = witness()
            r, cW2, cT = acc.fold(w, X)
            cW = oracle.curve_add(curve, cW, oracle.curve_mul(curve, tuple(from_limbs(cW2)), r))
            cE = oracle.curve_add(curve, cE, oracle.curve_mul(curve, tuple(from_limbs(cT)), r))
        assert acc.verify() == 0
        inst = acc.instance()
        assert tuple(from_limbs(inst["comm_W"])) == cW and tuple(from_limbs(inst["comm_E"])) == cE and inst["steps"] == 6
        zh, Eh = acc.z.download(), acc.E.download()
        bases = acc.ck.download(0, max(m, nrows))
        prods = [oracle.spmv(fid, nrows, ncols, *_csr(*M, nrows), zh) for M in (A, B, Cm)]
        assert oracle.first_unsat(fid, *prods, u=inst["u"], E=Eh) == -1
        assert oracle.msm(curve, bases[:m], zh[:m], threads=8) == cW
        assert oracle.msm(curve, bases[:nrows], Eh, threads=8) == cE
        # a flipped bit in E: relation and opening fail; in W: relation (some row), opening of W
        bad = Eh.copy(); bad[11, 0] ^= np.uint64(1)
        acc.E.upload(bad)
        assert acc.verify() & 5 == 5 and acc.shape.check_relaxed(acc.z, inst["u"], acc.E)[1] == 11
        acc.E.upload(Eh)
        badz = zh.copy(); badz[3, 0] ^= np.uint64(1)
        acc.z.upload(badz)
        assert acc.verify() & 2
        acc.z.upload(zh)
        assert acc.verify() == 0
    finally:
        acc.free()


def _circom_files(rng, q, n_in, n_mul, n_lin, n_witnesses):
    """The toy circuit as circom would emit it for the prime q — `.r1cs` bytes in circom's wire order [1 | outputs | private] and a few
    `.wtns` — written with the test-side iden3 writers."""
    from tests import _iden3
    nrows, ncols, m, A, B, Cm, witness = _toy_circuit(rng, q, n_in, n_mul, n_lin)
    n_pub = 2
    def wire(col):          # nova column -> circom wire
        return np.where(col == m, 0, np.where(col > m, col - m, col + n_pub + 1)).astype(np.uint32)
    vals_all = np.concatenate([M[2] for M in (A, B, Cm)])
    dict_canon, inv = np.unique(vals_all, axis=0, return_inverse=True)
    csr, off = [], 0
    for rows, cols, vals in (A, B, Cm):
        order = np.argsort(rows, kind="stable")
        rp = np.zeros(nrows + 1, dtype=np.uint32); np.add.at(rp, rows.astype(np.int64) + 1, 1)
        csr.append((np.cumsum(rp).astype(np.uint32), wire(cols.astype(np.int64))[order], inv.reshape(-1)[off:off + len(rows)][order].astype(np.uint32)))
        off += len(rows)
    r1cs_bytes = _iden3.write_r1cs(1 + n_pub + m, n_pub, 0, m, csr, dict_canon, prime=q)
    wtns = []
    for _ in range(n_witnesses):
        w, X = witness()
        wtns.append(_iden3.write_wtns(to_limbs([1] + X + w), prime=q))
    return r1cs_bytes, wtns, (nrows, m)


@pytest.mark.parametrize("curve", [_lib.CURVE_PALLAS, _lib.CURVE_VESTA])
def test_circom_artefacts_for_a_pasta_prime_fold_through_the_seam(ctx, oracle, curve):
    """`.r1cs` / `.wtns` as circom writes them for a Pasta prime (vimz_amd.iden3 reads them) folded with RelaxedAccumulator.from_r1cs —
    the nova-scotia flow (load_r1cs, one witness per step) on the cycle north_star names; verify() and the oracle accept."""
    from vimz_amd import iden3, nifs
    fid = _lib.CURVE_SCALAR_FIELD[curve]
    q = oracle.modulus[fid]
    rng = np.random.default_rng(3000 + curve)
    r1cs_bytes, wtns, (nrows, m) = _circom_files(rng, q, 16, 1200, 200, 4)
    r1cs = iden3.read_r1cs(r1cs_bytes)
    assert r1cs["prime"] == q and r1cs["n_constraints"] == nrows and r1cs["n_wires"] == m + 3
    acc = nifs.RelaxedAccumulator.from_r1cs(ctx, curve, r1cs)
    try:
        for blob in wtns:
            prime, vals = iden3.read_wtns(blob)
            assert prime == q
            acc.fold(*iden3.split_witness(r1cs, vals))
        assert acc.verify() == 0 and acc.steps == 4
        n_w, n_pub, A, B, Cm = iden3.to_nova_columns(r1cs)
        zh, Eh = acc.z.download(), acc.E.download()
        prods = [oracle.spmv(fid, nrows, n_w + 1 + n_pub, *_csr(*M, nrows), zh) for M in (A, B, Cm)]
        assert oracle.first_unsat(fid, *prods, u=acc.u, E=Eh) == -1
        assert oracle.msm(curve, acc.ck.download(0, n_w), zh[:n_w], threads=8) == tuple(from_limbs(acc.comm_W))
        with pytest.raises(ValueError):
            iden3.split_witness(r1cs, iden3.read_wtns(wtns[0])[1][:-1])
    finally:
        acc.free()


@pytest.mark.parametrize("fid", [_lib.FIELD_BN254_FR, _lib.FIELD_BN254_FQ, _lib.FIELD_PALLAS_FP, _lib.FIELD_VESTA_FQ])
def test_vec_axpy_matches_the_oracle(ctx, oracle, fid):
    """The fold of a resident vector (RelaxedR1CSWitness::fold) in all four fields: x1 + r·x2 on a prefix, the rest untouched."""
    from vimz_amd import hip
    q = oracle.modulus[fid]
    rng = np.random.default_rng(90 + fid)
    n = 10_000
    a = to_limbs([int.from_bytes(rng.bytes(32), "little") % q for _ in range(n)])
    b = to_limbs([int.from_bytes(rng.bytes(32), "little") % q for _ in range(n)])
    r = int.from_bytes(rng.bytes(16), "little") | 1 << 128
    x1, x2 = ctx.vec_from_host(fid, a), ctx.vec_from_host(fid, b)
    try:
        hip.vec_axpy(ctx, x1, r, x2, n=n - 7)
        got = x1.download()
        assert np.array_equal(got[: n - 7], oracle.axpy(fid, a[: n - 7], r, b[: n - 7]))
        assert np.array_equal(got[n - 7:], a[n - 7:])
        with pytest.raises(_lib.VimzError):
            hip.vec_axpy(ctx, x1, q, x2)                  # r not reduced
        with pytest.raises(_lib.VimzError):
            hip.vec_axpy(ctx, x1, 3, x1)
    finally:
        x1.free(); x2.free()


def test_upload_rejects_bad_shapes(ctx):
    from vimz_amd import hip
    one = to_limbs([1])
    rows, cols = np.array([0], dtype=np.uint32), np.array([0], dtype=np.uint32)
    ok = (rows, cols, one)
    with pytest.raises(_lib.VimzError):
        hip.R1CSShape(ctx, _lib.FIELD_BN254_FR, 1, 1, (np.array([1], dtype=np.uint32), cols, one), ok, ok)         # row outside the shape
    with pytest.raises(_lib.VimzError):
        hip.R1CSShape(ctx, _lib.FIELD_BN254_FR, 1, 1, ok, (rows, cols, np.full((1, 4), 2**64 - 1, dtype=np.uint64)), ok)   # coefficient >= p
    S = hip.R1CSShape(ctx, _lib.FIELD_BN254_FR, 1, 1, ok, ok, ok)
    z = ctx.vec_from_host(_lib.FIELD_BN254_FR, to_limbs([7]))
    az, bz, cz = S.multiply_vec(z)
    assert from_limbs(az.download()) == [7]
    for v in (az, bz, cz, z):
        v.free()
    S.free()
