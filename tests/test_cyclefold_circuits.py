"""Nova + CycleFold circuits on the host (no GPU): vimz_cf_selfcheck runs the recursion over the trivial step circuit with made-up
commitments and checks every witness against its R1CS, every in-circuit fold against field / curve arithmetic, and that no wire is left
unconstrained (every wire incremented by one violates a row)."""
from vimz_amd import hip


def test_cyclefold_circuits_selfcheck():
    res, counts = hip.cyclefold_selfcheck(5)
    assert res == 0, f"vimz_cf_selfcheck failed: bits {res:#x}"
    assert counts["cyclefold_constraints"] < 2000 and 20000 < counts["main_constraints"] < 40000      # one 128-bit scalar multiplication; four of them + 14 non-native folds + the hashes
    # no unconstrained wire: adding one to ANY wire of F' or of the CycleFold circuit violates a row that mentions it
    assert counts["main_flipped"] == counts["main_wires"] - 3 and counts["main_unnoticed"] == 0
    assert counts["cyclefold_flipped"] == counts["cyclefold_wires"] - 1 and counts["cyclefold_unnoticed"] == 0


def test_step_relation_of_the_main_circuit_restated_with_the_oracle(oracle):
    """The relation F' enforces, restated natively (tests/_cyclefold.py::step_relation: the oracle's Poseidon, Python integers, the oracle's
    arithmetic on both curves), applied to what the last step of the host-only run was given, returns what the circuit returned — base case,
    first fold into the zero instances, general steps.  No GPU."""
    import pytest
    from tests import _cyclefold as cfo
    for steps in (1, 2, 3, 6):
        dg, z0, words = hip.cyclefold_selfcheck_last_step(steps)
        s = cfo.parse_last_step(words, 1)
        assert s["i"] == steps - 1
        failed, Un, cn, x0, x1, (r, r1, r2) = cfo.step_relation(oracle, dg, [z0], s)
        assert failed == [], (steps, failed)
        assert (Un, cn, x0, x1) == (s["U_new"], s["cfU_new"], s["x0"], s["x1"]), steps
        mask = (1 << 128) - 1
        assert (r & mask, r1 & mask, r2 & mask) == (s["r"], s["r1"], s["r2"])


def test_merge_transcript_replayed_with_the_oracle(oracle):
    """The statement side of merged CycleFold proofs (segment hashes, adjacency, SHA3 transcript, folds of the instances on both curves, runs
    and their junctions): the library's replay of made-up, self-consistent records and the outside replay (tests/_cyclefold.py: hashlib,
    Python integers, the oracle's Poseidon and curve arithmetic) arrive at the same accumulator.  No GPU."""
    from tests import _cyclefold as cfo
    for run0, run1 in ((1, 0), (3, 0), (2, 2)):
        dg, rec_words, acc = hip.cyclefold_selfcheck_merge(run0, run1)
        rec = cfo.parse_merged_records(rec_words, 1)
        assert len(rec["segs"]) == run0 + run1 and len(rec["junctions"]) == (1 if run1 else 0)
        failed, got = cfo.replay_merged(oracle, rec, dg, 1)
        assert failed == [], failed
        assert got["n"] == acc["n"] and got["zs"] == acc["zs"] and got["ze"] == acc["ze"]
        assert [tuple(got["P"][0]), tuple(got["P"][1])] + list(got["P"][2:]) == [acc["P"][0], acc["P"][1]] + acc["P"][2:]
        assert [tuple(got["Q"][0]), tuple(got["Q"][1]), got["Q"][2], list(got["Q"][3])] == [acc["Q"][0], acc["Q"][1], acc["Q"][2], list(acc["Q"][3])]
        # a changed record is noticed by the outside replay
        bad = rec_words.copy(); bad[8 + len(rec["run_start"])] += 1             # n of the first segment
        f2, _ = cfo.replay_merged(oracle, cfo.parse_merged_records(bad, 1), dg, 1)
        assert f2


def test_helper_thread_wake_ups_are_not_lost():
    """The helper threads of the verifier circuits' witness generators spin for a while after a job and then sleep; posting a job to a helper that
    is just going to sleep must wake it (a release-ordered post could be passed by the read of the helper's `sleeping` flag: seen as a hang of a
    whole fold once the spin window was shortened).  200 000 post / wait pairs with the helper sleeping between all of them, in a process of its
    own with a time limit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes, sys; sys.path.insert(0, %r); from vimz_amd import _lib; L = _lib.testing_lib(); L.vimz_worker_selftest.restype = ctypes.c_int64; "
            "print(L.vimz_worker_selftest(200000))" % root)
    for spin in ("0", "3"):
        env = dict(os.environ, VIMZ_WORKER_SPIN_US=spin)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "200000", (spin, r.stdout[-200:], r.stderr[-300:])


def test_hash_outputs_are_decomposed_canonically():
    """the circuits derive their challenges from the bits of a hash output; circom-style Num2Bits(254) also accepts the bits of
    h + p when that is below 2^254 (about a third of all h), which would let a prover choose between two challenges.  bits_strict compares
    the bits with p - 1.  On the gadget alone, over both fields: every aliased witness satisfies the plain gadget's rows (the attack is
    real) and violates rows of the comparison and only those (the fix is what stops it); honest witnesses violate nothing — including
    h = 0, h = p - 1 and the largest aliasable h.  Inside F' (Nova + CycleFold over the trivial step circuit): every step run with aliased
    challenge decompositions is unsatisfiable.  Host only."""
    import ctypes
    import numpy as np
    from vimz_amd import _lib
    L = _lib.testing_lib()
    L.vimz_strict_bits_selfcheck.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for field in (0, 1):
        o = np.zeros(8, dtype=np.uint64)
        assert L.vimz_strict_bits_selfcheck(field, 64, o.ctypes.data) == 0
        tried, aliasable, plain_ok, only_strict_bad, honest_bad, steps, sat, steps_aliased = [int(x) for x in o]
        assert tried == 64 and 8 <= aliasable <= 40, o            # a third of random values, plus h = 0 and h = 2^254 - p - 1
        assert plain_ok == aliasable and only_strict_bad == aliasable and honest_bad == 0, o
        if field == 0:
            assert steps >= 3 and steps_aliased >= 1 and sat == 0, o


def test_decider_circuit_on_the_host():
    """The decider circuit (aug/decider.hpp) without a GPU: over the recursion of the trivial step circuit (the running witness folded on the host) its witness
    satisfies its R1CS, the public inputs are the reference contracts' layout (pp_hash, i, z_0, z_i, 4 x 5 limbs of the folded commitments, the KZG
    challenges and evaluations, 2 x 5 limbs of cmT: 36 + 2·len_z — contracts/ContrastVerifier.sol:700-772), a wrong KZG evaluation and a cmT other than the
    one the challenge was derived from are flagged, and every public input is noticed by some row (vimz_decider_selfcheck, testing library)."""
    import ctypes as C
    import numpy as np
    from vimz_amd import _lib
    T = _lib.testing_lib()
    T.vimz_decider_selfcheck.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_void_p]
    for steps in (2, 4):
        res, cnt = C.c_uint32(0xFFFF), np.zeros(4, dtype=np.uint64)
        assert T.vimz_decider_selfcheck(steps, 0, C.byref(res), cnt.ctypes.data) == 0
        assert res.value == 0, f"decider self-check bits {res.value:#x}"
        # one or two constraints per row of the main relation + the Horner chains + the hashes and bit decompositions
        assert int(cnt[2]) == 36 + 2 * 1 and int(cnt[3]) + 20000 < int(cnt[0]) < 4 * int(cnt[3]) + 30000


def test_full_decider_circuit_on_the_host():
    """The FULL decider (aug/decider_cf.hpp: what `DeciderEth` of vimz/src/sonobe_backend/decider.rs:13-21 attests beyond the `light-test` variant): the running
    CycleFold instance's two Pedersen commitments opened over Grumpkin and its relaxed relation checked over Fq in non-native limbs, inside the circuit.  Host
    only: three steps of the recursion with REAL CycleFold instances (witnesses of the CycleFold circuit for both folds of a step, committed and folded on
    the host; the running instance must equal the one F' folds in-circuit), then the decider's witness satisfies all 2.86 M rows with the same 38 public
    inputs; a changed CycleFold witness element, a changed error element, and a witness changed so that its commitment still opens are each flagged by the
    witness generator (-> VIMZ_ERR_UNSAT in vimz_decider_prove) and violate a row."""
    import ctypes as C
    import numpy as np
    from vimz_amd import _lib
    T = _lib.testing_lib()
    T.vimz_decider_selfcheck.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_void_p]
    res, cnt, light = C.c_uint32(0xFFFF), np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    assert T.vimz_decider_selfcheck(3, 1, C.byref(res), cnt.ctypes.data) == 0
    assert res.value == 0, f"full decider self-check bits {res.value:#x}"
    r2 = C.c_uint32(0xFFFF)
    assert T.vimz_decider_selfcheck(3, 0, C.byref(r2), light.ctypes.data) == 0
    # the same interface; 1 306 + 1 313 scalars at 3 rows a bit (2.0 M) + 1 313 rows at ~570 (0.75 M) more constraints, whatever the step circuit
    assert int(cnt[2]) == int(light[2]) == 38 and 2_700_000 < int(cnt[0]) - int(light[0]) < 2_800_000
