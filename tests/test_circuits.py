"""The native step-circuit builder (R1CS + witness program), checked on the CPU:
  * constraint / wire counts equal the reference's committed compile logs (circuit_parameters.csv),
  * the witness computed by the oracle's independent executor satisfies every R1CS row,
  * its public outputs equal the semantic step oracle (pinned on the reference's hashes/proofs),
  * tampered inputs are rejected."""
import numpy as np
import pytest

from tests import _data
from tests._oracle import (T_BLUR, T_BRIGHTNESS, T_CONTRAST, T_CROP, T_GRAYSCALE, T_HASH, T_REDACT, T_RESIZE, T_SHARPNESS,
                           from_limbs, r1cs_check, witness_execute)
from vimz_amd import image_editor as ie
from vimz_amd.circuit import Circuit

KAT = _data.kat()
ORC_T = {"blur": T_BLUR, "brightness": T_BRIGHTNESS, "contrast": T_CONTRAST, "grayscale": T_GRAYSCALE, "hash": T_HASH,
         "redact": T_REDACT, "resize": T_RESIZE, "sharpness": T_SHARPNESS}
OPS = list(ORC_T)


@pytest.fixture(scope="module")
def circuits():
    return {op: Circuit.for_resolution(op, "HD") for op in OPS}


@pytest.mark.parametrize("op", OPS)
def test_sizes_match_reference_compile_logs(circuits, op):
    c = circuits[op]
    ref = KAT["circuit_sizes"][op]
    assert c.n_constraints - c.n_linear == ref["constraints"]
    assert c.len_z == ref["public_inputs"] == ref["public_outputs"]
    assert c.n_priv == ref["private_inputs"]
    # circom drops private inputs no constraint touches (the 15 / 19 elements the truncating hasher never absorbs,
    # SURVEY.md F5); we keep them as wires with empty columns
    unused = {"hash": 15, "redact": 19}.get(op, 0)
    assert c.n_wires == ref["wires"] + unused


def test_width_scaling_model():
    # SURVEY.md Appendix A projections for the 4K / 8K configs
    c = Circuit.for_resolution("contrast", "4K")
    assert c.n_constraints - c.n_linear == 914592
    g = Circuit.for_resolution("grayscale", "4K")
    assert g.n_constraints == 361632
    r = Circuit.for_resolution("resize", "8K")
    assert r.n_constraints == 834480


def step_inputs(op, k=0):
    """(z0, per-step private inputs as (n,4) limbs) for steps 0..9 of the rows10 fixtures."""
    fx = _data.rows10(op)
    o = ie.hex_to_rows(fx["original"])
    tr = ie.hex_to_rows(fx["transformed"]) if "transformed" in fx else None
    if op == "hash":
        return [0], [o[i] for i in range(10)]
    if op in ("grayscale", "contrast", "brightness"):
        z0 = [0, 0] if op == "grayscale" else [0, 0, fx["factor"]]
        return z0, [np.concatenate([o[i], tr[i]]) for i in range(10)]
    if op in ("blur", "sharpness"):
        return [0, 0, 0, 0], [np.concatenate([o[i:i + 3].reshape(-1, 4), tr[i]]) for i in range(10)]
    if op == "resize":
        return [0, 0], [np.concatenate([o[3 * i:3 * i + 3].reshape(-1, 4), tr[2 * i:2 * i + 2].reshape(-1, 4)]) for i in range(10)]
    if op == "redact":
        red = [int(s, 16) for s in fx["redact"]]
        return [0, 0], [np.concatenate([o[i], np.array([[red[i], 0, 0, 0]], dtype=np.uint64)]) for i in range(10)]
    raise ValueError(op)


@pytest.mark.parametrize("op", OPS)
def test_witness_satisfies_r1cs_and_matches_step_semantics(oracle, circuits, op):
    c = circuits[op]
    z, inputs = step_inputs(op)
    kw = dict(width=160) if op == "redact" else {}
    for i in range(3):
        st, wires, z_out = witness_execute(oracle, c, z, inputs[i])
        assert st == 0
        assert r1cs_check(oracle, c, wires) == -1, f"{op}: step {i} witness violates the R1CS"
        ok, z_sem = oracle.step_eval(ORC_T[op], z, inputs[i], **kw)
        assert ok and z_out == z_sem
        # wire order: [1 | step_out | step_in | private inputs | ...]
        w = from_limbs(wires[:1 + 2 * c.len_z])
        assert w[0] == 1 and w[1:1 + c.len_z] == z_out and w[1 + c.len_z:] == z
        assert np.array_equal(wires[1 + 2 * c.len_z:1 + 2 * c.len_z + c.n_priv], inputs[i])
        z = z_out


@pytest.mark.parametrize("op", ["grayscale", "contrast", "sharpness", "resize"])
def test_tampered_row_is_rejected(oracle, circuits, op):
    c = circuits[op]
    z, inputs = step_inputs(op)
    bad = inputs[0].copy()
    bad[-1, 0] ^= np.uint64(0x40)     # flip a high bit of one channel of the last transformed element
    st, wires, _ = witness_execute(oracle, c, z, bad)
    assert st == 1 or r1cs_check(oracle, c, wires) != -1
    kw = {}
    ok, _ = oracle.step_eval(ORC_T[op], z, bad, **kw)
    assert not ok


def test_wrong_witness_wire_breaks_r1cs(oracle, circuits):
    c = circuits["hash"]
    z, inputs = step_inputs("hash")
    st, wires, _ = witness_execute(oracle, c, z, inputs[0])
    assert st == 0 and r1cs_check(oracle, c, wires) == -1
    wires2 = wires.copy()
    wires2[c.n_wires - 5, 0] ^= np.uint64(1)
    assert r1cs_check(oracle, c, wires2) != -1


def synthetic_resize_2to1(width=16, steps=3, seed=5):
    """Rows for the 2->1 resize relation used at 4K/8K (|a+b+c+d - 4t| <= 4): random pixels, exact box average."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(2 * steps, 10 * width, 3), dtype=np.uint8)
    small = ((img[0::2, 0::2].astype(np.int64) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2]) // 4).astype(np.uint8)
    o, t = ie.compress_by_rows(img), ie.compress_by_rows(small)
    return [0, 0], [np.concatenate([o[2 * i:2 * i + 2].reshape(-1, 4), t[i]]) for i in range(steps)]


def test_resize_2to1_extension(oracle):
    """Our extension for the 4K/8K configs (the reference only instantiates 3->2, resize_step.circom:81): builder, oracle
    executor and semantic oracle agree, and a wrong average is rejected."""
    c = Circuit("resize", 16, 8, 2, 1, 0)
    z, inputs = synthetic_resize_2to1()
    for i in range(3):
        st, wires, z_out = witness_execute(oracle, c, z, inputs[i])
        assert st == 0 and r1cs_check(oracle, c, wires) == -1
        ok, z_sem = oracle.step_eval(T_RESIZE, z, inputs[i], width=16, width2=8, rows_in=2, rows_out=1)
        assert ok and z_sem == z_out
        z = z_out
    bad = inputs[0].copy()
    bad[-1, 0] ^= np.uint64(0x20)
    st, wires, _ = witness_execute(oracle, c, [0, 0], bad)
    assert st == 1 or r1cs_check(oracle, c, wires) != -1
    assert not oracle.step_eval(T_RESIZE, [0, 0], bad, width=16, width2=8, rows_in=2, rows_out=1)[0]


@pytest.fixture(scope="module")
def crop_circuit():
    return Circuit.for_resolution("crop", "HD")


def test_crop_sizes_match_reference_compile_log(crop_circuit):
    """BASELINE config #1 (crop_step at HD): the builder reproduces circom's counts exactly
    (circuits/nova_snark/circuit_parameters.csv:5: 672 272 non-linear + 1 linear constraints, 671 633 wires)."""
    c, ref = crop_circuit, KAT["circuit_sizes"]["crop"]
    assert c.n_constraints - c.n_linear == ref["constraints"] == 672272 and c.n_linear == 1
    assert c.n_wires == ref["wires"] == 671633
    assert c.len_z == 3 and c.n_priv == ref["private_inputs"] == 128


def test_crop_witness_and_semantics(oracle, crop_circuit):
    """Crop followed literally from the Circom text (x = info bits 0-11 and `info + 1` advances x): the
    executor's witness satisfies the R1CS and the state equals the oracle's literal restatement, inside and outside the
    row window."""
    c = crop_circuit
    fx = _data.rows10("crop")
    o = ie.hex_to_rows(fx["original"])
    for info in (fx["info"], (150 << 24) | (100 << 12) | 37):      # row_index 200 (outside y..y+480? 100<=200<580: inside), then x = 37
        z = [0, 0, info]
        for i in range(2):
            st, wires, z_out = witness_execute(oracle, c, z, o[i])
            assert st == 0
            assert r1cs_check(oracle, c, wires) == -1
            ok, z_sem = oracle.step_eval(T_CROP, z, o[i], width=128, width2=64, crop_h=480)
            assert ok and z_sem == z_out and z_out[2] == z[2] + 1
            z = z_out
    # a row index below the window leaves the cropped hash untouched
    z = [5, 7, (50 << 24) | (100 << 12)]
    st, wires, z_out = witness_execute(oracle, c, z, o[0])
    assert st == 0 and r1cs_check(oracle, c, wires) == -1 and z_out[1] == 7
    # x beyond the row: the Decoder has no solution
    z = [0, 0, (200 << 24) | (100 << 12) | 1280]
    st, wires, _ = witness_execute(oracle, c, z, o[0])
    assert st == 1 or r1cs_check(oracle, c, wires) != -1
    assert not oracle.step_eval(T_CROP, z, o[0], width=128, width2=64, crop_h=480)[0]
