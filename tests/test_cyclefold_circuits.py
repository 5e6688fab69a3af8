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
