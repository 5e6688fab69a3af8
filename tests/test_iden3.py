"""iden3 `.r1cs` / `.wtns` readers: a circuit written out in circom's binary format and read back by the
native loader is the same R1CS, and accepts the same witnesses."""
import numpy as np
import pytest

from tests import _iden3
from tests._oracle import r1cs_check, witness_execute
from tests.test_circuits import step_inputs
from vimz_amd.circuit import Circuit, wtns_load


@pytest.fixture(scope="module")
def hash_circuit():
    return Circuit.for_resolution("hash", "HD")


def _to_r1cs_bytes(c):
    csr = [c.csr(m) for m in "ABC"]
    return _iden3.write_r1cs(c.n_wires, c.len_z, c.len_z, c.n_priv, csr, c.export("DICT_CANON", np.uint64).reshape(-1, 4))


def test_r1cs_round_trip(oracle, hash_circuit):
    c = hash_circuit
    loaded = Circuit.from_r1cs(_to_r1cs_bytes(c))
    assert (loaded.n_wires, loaded.n_constraints, loaded.len_z, loaded.n_priv) == (c.n_wires, c.n_constraints, c.len_z, c.n_priv)
    assert (loaded.nnz_a, loaded.nnz_b, loaded.nnz_c) == (c.nnz_a, c.nnz_b, c.nnz_c)
    for m in "ABC":
        a, b = c.csr(m), loaded.csr(m)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        da, db = c.export("DICT_CANON", np.uint64).reshape(-1, 4), loaded.export("DICT_CANON", np.uint64).reshape(-1, 4)
        assert np.array_equal(da[a[2]], db[b[2]])                      # same coefficients, dictionary order may differ
    z0, inputs = step_inputs("hash")
    st, wires, _ = witness_execute(oracle, c, z0, inputs[0])
    assert r1cs_check(oracle, loaded, wires) == -1
    bad = wires.copy(); bad[-3, 0] ^= np.uint64(1)
    assert r1cs_check(oracle, loaded, bad) != -1


def test_wtns_round_trip(oracle, hash_circuit):
    z0, inputs = step_inputs("hash")
    _, wires, _ = witness_execute(oracle, hash_circuit, z0, inputs[0])
    assert np.array_equal(wtns_load(_iden3.write_wtns(wires)), wires)


def test_malformed_files_are_rejected():
    from vimz_amd import _lib
    with pytest.raises(_lib.VimzError):
        Circuit.from_r1cs(b"r1cs" + b"\\x00" * 40)
    with pytest.raises(_lib.VimzError):
        wtns_load(b"nope")


@pytest.mark.gpu
def test_fold_external_witnesses_of_loaded_r1cs(oracle, hash_circuit):
    """The drop-in seam for circom artefacts: .r1cs loaded, .wtns witnesses folded on the GPU, accumulator verifies."""
    from vimz_amd import _lib, hip
    from tests._oracle import from_limbs
    c = hash_circuit
    loaded = Circuit.from_r1cs(_to_r1cs_bytes(c))
    z0, inputs = step_inputs("hash")
    z, wits = list(z0), []
    for i in range(5):
        _, w, z = witness_execute(oracle, c, z, inputs[i])
        wits.append(wtns_load(_iden3.write_wtns(w)))
    ctx = hip.Context(0)
    ck = ctx.bases_generate(_lib.CURVE_BN254_G1, 1 << 13)
    P = hip.Prover(ctx, loaded, ck, max_batch=2)
    try:
        P.reset(z0)
        P.fold_witness(np.stack(wits))
        assert P.verify() == 0
        inst = P.instance()
        assert inst["steps"] == 5 and from_limbs(inst["z"]) == z
        # a witness that does not continue the chain is refused
        P.reset(z0)
        with pytest.raises(_lib.VimzError):
            P.fold_witness(np.stack([wits[1]]))
        # the native-program path refuses a loaded circuit
        with pytest.raises(_lib.VimzError):
            P.fold(np.stack(inputs[:1]))
    finally:
        P.close(); ck.free(); ctx.close()


@pytest.mark.gpu
def test_ivc_over_a_loaded_r1cs_with_external_witnesses(oracle, hash_circuit):
    """The same seam in IVC mode: a circom-style .r1cs gets Nova's verifier circuit appended (vimz_ivc_create works on any step
    circuit in the [1 | out | in | ...] layout), its .wtns witnesses are folded with vimz_ivc_fold_witness, the proof verifies."""
    from vimz_amd import _lib, hip
    c = hash_circuit
    loaded = Circuit.from_r1cs(_to_r1cs_bytes(c))
    z0, inputs = step_inputs("hash")
    z, wits = list(z0), []
    for i in range(5):
        _, w, z = witness_execute(oracle, c, z, inputs[i])
        wits.append(wtns_load(_iden3.write_wtns(w)))
    ctx = hip.Context(0)
    ck1 = ctx.bases_generate(_lib.CURVE_BN254_G1, 1 << 14)
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-secondary")
    ivc = hip.IVC(ctx, loaded, ck1, ck2, max_batch=2)
    try:
        ivc.reset(z0)
        ivc.fold_witness(np.stack(wits))
        assert ivc.verify(5, z0) == 0 and ivc.state() == (z, 5)
        with pytest.raises(_lib.VimzError):
            ivc.fold(np.stack(inputs[:1]))               # a loaded circuit has no witness program
        with pytest.raises(_lib.VimzError):
            ivc.fold_witness(np.stack([wits[0]]))        # does not continue the chain
    finally:
        ivc.close(); ck1.free(); ck2.free(); ctx.close()


def test_python_reader_handles_any_prime():
    """vimz_amd.iden3 (the reader that feeds the seam-based accumulator) on files written for the Pallas base field: header, triplets in
    circom's wire numbering, the move to nova-snark's column order, witnesses."""
    import numpy as np
    from tests import _iden3
    from tests._oracle import Q_VESTA, to_limbs
    from vimz_amd import iden3
    # two constraints over wires [1 | out | in | a, b]:  a*b = out ;  (in + 3)*1 = a
    dict_canon = to_limbs([1, 3, Q_VESTA - 1])
    A = (np.array([0, 1, 3], dtype=np.uint32), np.array([3, 2, 0], dtype=np.uint32), np.array([0, 0, 1], dtype=np.uint32))
    B = (np.array([0, 1, 2], dtype=np.uint32), np.array([4, 0], dtype=np.uint32), np.array([0, 0], dtype=np.uint32))
    C = (np.array([0, 1, 2], dtype=np.uint32), np.array([1, 3], dtype=np.uint32), np.array([0, 0], dtype=np.uint32))
    blob = _iden3.write_r1cs(5, 1, 1, 2, [A, B, C], dict_canon, prime=Q_VESTA)
    r = iden3.read_r1cs(blob)
    assert (r["prime"], r["n_wires"], r["n_pub_out"], r["n_pub_in"], r["n_prv"], r["n_constraints"]) == (Q_VESTA, 5, 1, 1, 2, 2)
    assert r["A"][0].tolist() == [0, 1, 1] and r["A"][1].tolist() == [3, 2, 0] and r["A"][2][2].tolist() == [3, 0, 0, 0]
    n_w, n_pub, An, Bn, Cn = iden3.to_nova_columns(r)
    assert (n_w, n_pub) == (2, 2)
    assert An[1].tolist() == [0, 4, 2] and Bn[1].tolist() == [1, 2] and Cn[1].tolist() == [3, 0]      # W = [a, b], u at 2, X = [out, in] at 3, 4
    wt = _iden3.write_wtns(to_limbs([1, 35, 4, 7, 5]), prime=Q_VESTA)
    prime, vals = iden3.read_wtns(wt)
    assert prime == Q_VESTA
    W, X = iden3.split_witness(r, vals)
    assert W[:, 0].tolist() == [7, 5] and X == [35, 4]
