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
