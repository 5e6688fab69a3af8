"""Self-consistency of the oracle's field / curve / MSM / relaxed-R1CS algebra against Python big-int
arithmetic (no reference vectors exist for these; SURVEY.md §8c last paragraph)."""
import random

import numpy as np
import pytest

from tests._oracle import CURVE_SCALAR, GENERATORS, from_limbs, to_limbs

MODULI = {
    0: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
    1: 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
    2: 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,
    3: 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001,
}


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_field_ops(oracle, fid):
    p = MODULI[fid]
    assert oracle.modulus[fid] == p
    rng = random.Random(fid)
    vals = [0, 1, 2, p - 1, p - 2, (1 << 255) % p, (1 << 128) - 1] + [rng.randrange(p) for _ in range(40)]
    for a in vals:
        for b in vals[:12]:
            assert oracle.f_add(fid, a, b) == (a + b) % p
            assert oracle.f_sub(fid, a, b) == (a - b) % p
            assert oracle.f_mul(fid, a, b) == (a * b) % p
        if a:
            assert oracle.f_inv(fid, a) == pow(a, p - 2, p)
    lim = to_limbs(vals)
    m = oracle.to_mont(fid, lim)
    assert from_limbs(m) == [(v << 256) % p for v in vals]
    assert from_limbs(oracle.from_mont(fid, m)) == vals


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_curve_group_law(oracle, cid):
    G = GENERATORS[cid]
    r = MODULI[CURVE_SCALAR[cid]]
    assert oracle.on_curve(cid, G)
    assert oracle.curve_mul(cid, G, r) == (0, 0)            # group order = scalar-field modulus
    assert oracle.curve_mul(cid, G, r + 1) == G
    rng = random.Random(cid)
    a, b = rng.randrange(r), rng.randrange(r)
    A, B = oracle.curve_mul(cid, G, a), oracle.curve_mul(cid, G, b)
    assert oracle.on_curve(cid, A) and oracle.on_curve(cid, B)
    assert oracle.curve_add(cid, A, B) == oracle.curve_mul(cid, G, (a + b) % r)
    assert oracle.curve_add(cid, A, A) == oracle.curve_mul(cid, G, (2 * a) % r)    # doubling branch
    negA = oracle.curve_mul(cid, G, r - a)
    assert oracle.curve_add(cid, A, negA) == (0, 0)                               # inverse branch
    assert oracle.curve_add(cid, A, (0, 0)) == A and oracle.curve_add(cid, (0, 0), B) == B


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
@pytest.mark.parametrize("n", [1, 7, 300, 3000])
def test_msm_equals_discrete_log_sum(oracle, cid, n):
    """bases P_i = (i+1)·G so MSM(s, P) must equal (Σ s_i (i+1) mod r)·G."""
    r = MODULI[CURVE_SCALAR[cid]]
    rng = random.Random(n * 4 + cid)
    bases = oracle.seq_bases(cid, n)
    scal = [rng.randrange(r) for _ in range(n)]
    if n >= 7:
        scal[0] = 0; scal[1] = 1; scal[2] = r - 1; scal[3] = 255
    k = sum(s * (i + 1) for i, s in enumerate(scal)) % r
    want = oracle.curve_mul(cid, GENERATORS[cid], k)
    assert oracle.msm(cid, bases, to_limbs(scal), threads=1) == want
    assert oracle.msm(cid, bases, to_limbs(scal), threads=4) == want


def test_msm_identity_and_repeated_bases(oracle):
    cid = 0
    G = GENERATORS[cid]
    r = MODULI[0]
    n = 64
    bases = np.tile(to_limbs(G).reshape(1, 8), (n, 1))
    bases[5] = 0                                                # identity base
    scal = [3] * n
    want = oracle.curve_mul(cid, G, (3 * (n - 1)) % r)
    assert oracle.msm(cid, bases, to_limbs(scal), threads=1) == want
    assert oracle.msm(cid, bases[:0], to_limbs([]), threads=1) == (0, 0)


def test_relaxed_r1cs_fold_identity(oracle):
    """Folding two satisfying instances with the cross term keeps Az∘Bz = u·Cz + E (Nova §4)."""
    p = MODULI[0]
    rng = random.Random(7)
    nrows, ncols = 50, 90
    # random sparse A,B; C chosen per instance is impossible for one shape, so build z's that satisfy by
    # constructing C rows with a single entry on a dedicated column whose z value we solve for.
    def rand_csr(avoid_last):
        rp, col, val = [0], [], []
        for r_ in range(nrows):
            k = rng.randrange(1, 5)
            cols = rng.sample(range(ncols - nrows), k)
            col += cols; val += [rng.randrange(p) for _ in cols]
            rp.append(len(col))
        return np.array(rp, dtype=np.uint32), np.array(col, dtype=np.uint32), to_limbs(val)
    A = rand_csr(True); B = rand_csr(True)
    # C: row i has the single entry 1 at column (ncols - nrows + i)
    C_ = (np.arange(nrows + 1, dtype=np.uint32), np.arange(ncols - nrows, ncols, dtype=np.uint32), to_limbs([1] * nrows))

    def make_z():
        z = [1] + [rng.randrange(p) for _ in range(ncols - nrows - 1)] + [0] * nrows
        zl = to_limbs(z)
        az = from_limbs(oracle.spmv(0, nrows, ncols, *A, zl)); bz = from_limbs(oracle.spmv(0, nrows, ncols, *B, zl))
        for i in range(nrows):
            z[ncols - nrows + i] = az[i] * bz[i] % p
        return z
    z1, z2 = make_z(), make_z()
    def abc(z):
        zl = to_limbs(z)
        return [oracle.spmv(0, nrows, ncols, *M, zl, threads=2) for M in (A, B, C_)]
    a1, b1, c1 = abc(z1); a2, b2, c2 = abc(z2)
    assert oracle.first_unsat(0, a1, b1, c1) == -1 and oracle.first_unsat(0, a2, b2, c2) == -1
    T = oracle.cross_term(0, a1, b1, c1, 1, a2, b2, c2, 1)
    rch = rng.randrange(1 << 128)
    zf = [(x + rch * y) % p for x, y in zip(z1, z2)]          # u folds too: zf[0] = 1 + r
    E = oracle.axpy(0, to_limbs([0] * nrows), rch, T)
    af, bf, cf = abc(zf)
    assert oracle.first_unsat(0, af, bf, cf, u=zf[0], E=E) == -1
    Ebad = E.copy(); Ebad[3, 0] ^= np.uint64(1)
    assert oracle.first_unsat(0, af, bf, cf, u=zf[0], E=Ebad) == 3
    # linearity used by the GPU design: A(z1 + r z2) = Az1 + r Az2
    assert from_limbs(af) == from_limbs(oracle.axpy(0, a1, rch, a2))
