#!/usr/bin/env python3
"""Verify serialised IVC proofs in a process of their own (the `verify_folded_proof(proof, params, …)` side of the reference,
vimz/src/nova_snark_backend/folding.rs:45-56): rebuilds the public parameters (step circuit, both commitment keys) from the
transformation name and resolution, imports each proof blob and runs vimz_ivc_verify; for several segment proofs of one image
it also checks that the boundary states chain.  The statement verified is the reference's: iteration_count(transformation,
resolution) steps starting from Transformation::ivc_initial_state (factor / info given on the command line).
usage: verify_proof.py <transformation> <resolution> <factor-or-info> <proof.bin> [<proof.bin> ...]
(write proofs with: tools/e2e.py <transformation> <resolution> <segments> ivc <prefix>  ->  <prefix>.<k>.bin)"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from vimz_amd import _lib, folding, hip  # noqa: E402,F401


def main():
    t, res, extra, paths = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4:]
    z_first = folding.ivc_initial_state(t, {"factor": extra, "info": extra})
    want_steps = folding.iteration_count(t, res)
    ctx = hip.Context(0)
    circuit, params = folding.prepare_folding(ctx, t, res)
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-secondary")
    ivc = hip.IVC(ctx, circuit, params.ck, ck2, max_batch=1)
    ok, prev_end, total = True, None, 0
    for p in paths:
        ivc.proof_import(np.fromfile(p, dtype=np.uint8))
        z_end, steps = ivc.state()
        # the statement comes from the verifier, never from the blob: this segment must start where the previous one ended
        # (the first one at the transformation's initial state) — vimz_ivc_verify checks the proof against exactly that
        z_claim = z_first if prev_end is None else prev_end
        code = ivc.verify(steps, z_claim)
        print(f"{p}: {steps} steps, verify code {code} against the start state {'z0' if prev_end is None else 'the previous proof ended in'}")
        ok = ok and code == 0
        prev_end, total = z_end, total + steps
    if total != want_steps:
        print(f"step count {total} differs from iteration_count({t}, {res}) = {want_steps}")
        ok = False
    print("ACCEPTED" if ok else "REJECTED", f"({total} steps, final state {[hex(z) for z in prev_end]})")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
