#!/usr/bin/env python3
"""Flip every wire of a verifier-circuit witness in turn and require the R1CS (checked by the CPU oracle) to break:
no wire of the augmented circuits is unconstrained.  CPU only, ~45 s per side.  usage: aug_exhaustive.py"""
import random
import sys
import time

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from tests import _oracle  # noqa: E402
from tests._oracle import from_limbs, to_limbs  # noqa: E402
from tests.test_augmented import SIDES, _case  # noqa: E402
from vimz_amd import hip  # noqa: E402

orc = _oracle.load()
for side in (0, 1):
    c = hip.AugCircuit(side)
    tabs = c.r1cs()
    fid = SIDES[side][0]
    p = orc.modulus[fid]
    pz, i, z, U, u, T = _case(orc, side, "generic", random.Random(3))
    wires, out = c.witness([pz, i, z, *U, *u, *T])
    assert orc.r1cs_check_relaxed(fid, tabs, c.n_wires, wires) == -1
    t0, free = time.time(), []
    for k in range(1, c.n_wires):
        w = wires.copy()
        w[k] = to_limbs([(from_limbs(w[k:k + 1])[0] + 1) % p])[0]
        if orc.r1cs_check_relaxed(fid, tabs, c.n_wires, w, threads=2) < 0:
            free.append(k)
    print(f"side {side}: {len(free)} unconstrained wires of {c.n_wires} ({time.time() - t0:.1f} s)", free[:20])
