#!/usr/bin/env python3
"""Fold the ten fixture rows of a step circuit (several times over) as one Nova IVC proof and print a digest of everything the
proof consists of (both running instances, the last fresh secondary instance, the state), plus the verification code.
The proof is a deterministic function of the inputs, whatever the schedule: tests/test_gpu_ivc.py runs this under the
library's debugging switches (serial streams, no fused small MSM, no window tables, ...) and compares the lines.
usage: ivc_digest.py [transformation] [repeats] [max_batch]"""
import hashlib
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from tests.test_circuits import step_inputs  # noqa: E402
from vimz_amd import _lib as L, hip  # noqa: E402
from vimz_amd.circuit import Circuit  # noqa: E402


def main():
    t = sys.argv[1] if len(sys.argv) > 1 else "hash"
    rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    max_batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    z0, inputs = step_inputs(t)
    rows = np.concatenate([np.stack(inputs)] * rep)
    ctx = hip.Context(0)
    c = Circuit.for_resolution(t, "HD")
    n = 1
    while n < max(c.n_wires, c.n_constraints) + 8192:
        n *= 2
    ck = ctx.bases_generate(L.CURVE_BN254_G1, n)
    import os
    if int(os.environ.get("VIMZ_DIGEST_TABLES", "0")):      # window tables of the key (the large MSM's default in folding.prepare_folding): same proof
        ck.precompute(int(os.environ["VIMZ_DIGEST_TABLES"]))
    ck2 = ctx.bases_generate(L.CURVE_GRUMPKIN, 8192, b"ck-secondary")
    ivc = hip.IVC(ctx, c, ck, ck2, max_batch=max_batch)
    ivc.reset(z0)
    ivc.fold(rows[:7])          # two calls: the second starts without a queued large MSM
    ivc.fold(rows[7:])
    code = ivc.verify(len(rows), z0)
    h = hashlib.sha256()
    for side in (0, 1):
        h.update(np.ascontiguousarray(ivc.export(side, hip.IX_INSTANCE)).tobytes())
    h.update(np.ascontiguousarray(ivc.export(1, hip.IX_FRESH_INSTANCE)).tobytes())
    z, steps = ivc.state()
    h.update(repr((z, steps)).encode())
    print(f"digest {h.hexdigest()} verify {code} steps {steps}")
    ivc.close()
    ctx.close()


if __name__ == "__main__":
    main()
