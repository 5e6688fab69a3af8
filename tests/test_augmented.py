"""The augmented verifier circuits (vimz_amd/csrc/aug/) on their own, without a GPU:
the R1CS the product builds and the witness its generator computes are checked by the oracle (oracle/nova.hpp: the
relation stated natively; generic relaxed-R1CS check), on both sides of the BN254/Grumpkin cycle."""
import random

import numpy as np
import pytest

from tests._oracle import GENERATORS, from_limbs, to_limbs

# side -> (field id of the circuit, curve id of the commitments it folds, field id of the folded instances' scalars)
SIDES = {0: (0, 1, 1), 1: (1, 0, 0)}


@pytest.fixture(scope="module")
def circuits():
    from vimz_amd import hip
    cs = {s: hip.AugCircuit(s) for s in SIDES}
    tabs = {s: cs[s].r1cs() for s in SIDES}
    yield cs, tabs
    for c in cs.values():
        c.close()


def _pt(oracle, cid, k):
    return (0, 0) if k == 0 else oracle.curve_mul(cid, GENERATORS[cid], k)


def _case(oracle, side, name, rng):
    fid, cid, gid = SIDES[side]
    p_other = oracle.modulus[gid]
    pz, z = rng.randrange(1 << 250), rng.randrange(1 << 200)       # pz: the shape digest the hashes absorb
    z0 = z if name == "base" else rng.randrange(1 << 200)
    r250 = lambda: rng.randrange(1 << 250)
    if name == "base":
        i = 0
        U = [*_pt(oracle, cid, 11), *_pt(oracle, cid, 12), 77, rng.randrange(p_other), rng.randrange(p_other)]   # ignored by the base case
        u = [*_pt(oracle, cid, 5), r250(), r250()]
        T = (0, 0)
    elif name == "first_fold":      # running instance still has E = identity (after one fold from the zero instance)
        i = 1
        U = [*_pt(oracle, cid, 9), 0, 0, (1 << 128) + 12345, rng.randrange(p_other), rng.randrange(p_other)]
        u = [*_pt(oracle, cid, 6), 0, r250()]
        T = _pt(oracle, cid, 31)
    elif name == "zero_running":    # the primary at step 1: U is still the zero instance, T is the identity
        i = 1
        U = [0, 0, 0, 0, 0, 0, 0]
        u = [*_pt(oracle, cid, 8), 0, r250()]
        T = (0, 0)
    else:
        i = rng.randrange(2, 1 << 20)
        U = [*_pt(oracle, cid, rng.randrange(1, 1 << 64)), *_pt(oracle, cid, rng.randrange(1, 1 << 64)), rng.randrange(1 << 140), rng.randrange(p_other), rng.randrange(p_other)]
        u = [*_pt(oracle, cid, rng.randrange(1, 1 << 64)), 0, r250()]
        T = _pt(oracle, cid, rng.randrange(1, 1 << 64))
    if i > 0:
        u[2] = oracle.nova_instance_hash(fid, pz, i, [z0], [z], U)
    return pz, i, z0, z, U, u, T


@pytest.mark.parametrize("side", [0, 1])
def test_verifier_circuit_size(circuits, side):
    cs, tabs = circuits
    c = cs[side]
    assert 7000 < c.n_constraints < 9000 and c.n_wires - 3 <= c.n_constraints + 16
    t = tabs[side]
    assert len(t["A_rowptr"]) == c.n_constraints + 1 and int(t["A_col"].max()) < c.n_wires


@pytest.mark.parametrize("side", [0, 1])
@pytest.mark.parametrize("name", ["base", "first_fold", "zero_running", "generic", "generic2"])
def test_witness_satisfies_r1cs_and_equals_native_relation(oracle, circuits, side, name):
    cs, tabs = circuits
    c = cs[side]
    fid = SIDES[side][0]
    rng = random.Random(f"{side}-{name}")
    pz, i, z0, z, U, u, T = _case(oracle, side, name, rng)
    wires, out = c.witness([pz, i, z0, z, *U, *u, *T])
    assert out[10] == 0, "range flag"
    assert oracle.r1cs_check_relaxed(fid, tabs[side], c.n_wires, wires) == -1
    want = oracle.nova_step(side, side == 0, pz, i, [z0], [z], [z], U, u, T)
    assert want is not None
    U_new, rho, x1 = want
    assert out[0:7] == U_new and out[7] == rho and out[9] == x1 and out[8] == u[3]
    assert rho >> 128 == 1
    # the public IO sits in the last two wires
    assert from_limbs(wires[-2:]) == [u[3], x1]


@pytest.mark.parametrize("side", [0, 1])
def test_wrong_incoming_hash_is_rejected(oracle, circuits, side):
    cs, tabs = circuits
    c = cs[side]
    fid = SIDES[side][0]
    pz, i, z0, z, U, u, T = _case(oracle, side, "generic", random.Random(7))
    u[2] ^= 1
    assert oracle.nova_step(side, side == 0, pz, i, [z0], [z], [z], U, u, T) is None
    wires, out = c.witness([pz, i, z0, z, *U, *u, *T])
    assert out[10] == 1
    assert oracle.r1cs_check_relaxed(fid, tabs[side], c.n_wires, wires) >= 0


@pytest.mark.parametrize("side", [0, 1])
def test_base_case_must_start_from_z0(oracle, circuits, side):
    """A chain that claims z_0 but feeds the first step another state is not satisfiable (without this the IVC
    proved only that SOME start state leads to z_n), and z_0 reaches every later hash: changing it changes the output hash."""
    cs, tabs = circuits
    c = cs[side]
    fid = SIDES[side][0]
    pz, i, z0, z, U, u, T = _case(oracle, side, "base", random.Random(3))
    assert i == 0 and z0 == z
    wires, out = c.witness([pz, i, z0, z, *U, *u, *T])
    assert out[10] == 0 and oracle.r1cs_check_relaxed(fid, tabs[side], c.n_wires, wires) == -1
    x1_good = out[9]
    assert oracle.nova_step(side, side == 0, pz, 0, [z0 + 1], [z], [z], U, u, T) is None
    wires, out = c.witness([pz, i, z0 + 1, z, *U, *u, *T])
    assert out[10] == 1
    assert oracle.r1cs_check_relaxed(fid, tabs[side], c.n_wires, wires) >= 0
    # later steps: z_0 is free to differ from z_i but is bound by the hashes
    pz, i, z0, z, U, u, T = _case(oracle, side, "generic", random.Random(4))
    _, out_a = c.witness([pz, i, z0, z, *U, *u, *T])
    u_b = list(u); u_b[2] = oracle.nova_instance_hash(fid, pz, i, [z0 + 1], [z], U)
    _, out_b = c.witness([pz, i, z0 + 1, z, *U, *u_b, *T])
    assert out_a[10] == 0 and out_b[10] == 0 and out_a[9] != out_b[9] and x1_good != out_a[9]
    wires, out = c.witness([pz, i, z0 + 1, z, *U, *u, *T])          # the old hash with another z_0: rejected
    assert out[10] == 1 and oracle.r1cs_check_relaxed(fid, tabs[side], c.n_wires, wires) >= 0


@pytest.mark.parametrize("side", [0, 1])
def test_tampered_witness_and_off_curve_points_are_rejected(oracle, circuits, side):
    cs, tabs = circuits
    c = cs[side]
    fid = SIDES[side][0]
    p = oracle.modulus[fid]
    rng = random.Random(11 + side)
    pz, i, z0, z, U, u, T = _case(oracle, side, "generic", rng)
    wires, out = c.witness([pz, i, z0, z, *U, *u, *T])
    assert oracle.r1cs_check_relaxed(fid, tabs[side], c.n_wires, wires) == -1
    for k in rng.sample(range(1, c.n_wires), 1500):
        w = wires.copy()
        w[k] = to_limbs([(from_limbs(w[k:k + 1])[0] + 1) % p])[0]
        assert oracle.r1cs_check_relaxed(fid, tabs[side], c.n_wires, w, threads=2) >= 0, f"wire {k} is unconstrained"
    # a fresh commitment that is not on the curve
    bad_u = list(u); bad_u[1] = (bad_u[1] + 1) % p
    bad_u[2] = u[2]
    wires, out = c.witness([pz, i, z0, z, *U, *bad_u, *T])
    assert oracle.r1cs_check_relaxed(fid, tabs[side], c.n_wires, wires) >= 0
    bad_T = (T[0], (T[1] + 1) % p)
    wires, out = c.witness([pz, i, z0, z, *U, *u, *bad_T])
    assert oracle.r1cs_check_relaxed(fid, tabs[side], c.n_wires, wires) >= 0


def test_poseidon_over_fq_differs_from_fr_and_is_shared_with_the_oracle(oracle):
    ins = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]
    a, b = oracle.nova_hash(0, ins), oracle.nova_hash(1, ins)
    assert a != b
    assert oracle.nova_hash(0, [1, 2]) == oracle.poseidon([1, 2])   # the Fr instance is circomlib's (pinned KAT)
