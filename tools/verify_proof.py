#!/usr/bin/env python3
"""Verify serialised IVC proofs in a process of their own (the `verify_folded_proof(proof, params, …)` side of the reference,
vimz/src/nova_snark_backend/folding.rs:45-56): rebuilds the public parameters (step circuit, both commitment keys) from the
transformation name and resolution, imports each proof blob and runs vimz_ivc_verify; for several segment proofs of one image
it also checks that the boundary states chain.
usage: verify_proof.py <transformation> <resolution> <proof.bin> [<proof.bin> ...]
(write proofs with: tools/e2e.py <transformation> <resolution> <segments> ivc <prefix>  ->  <prefix>.<k>.bin)"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from vimz_amd import _lib, folding, hip  # noqa: E402


def main():
    t, res, paths = sys.argv[1], sys.argv[2], sys.argv[3:]
    ctx = hip.Context(0)
    circuit, params = folding.prepare_folding(ctx, t, res)
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-secondary")
    ivc = hip.IVC(ctx, circuit, params.ck, ck2, max_batch=1)
    ok, prev_end, total = True, None, 0
    for p in paths:
        ivc.proof_import(np.fromfile(p, dtype=np.uint8))
        code = ivc.verify()
        z_end, steps = ivc.state()
        z0 = hip._export(ctx.lib.vimz_ivc_export, ivc.h, 0, hip.IX_PARAMS).view(np.uint64).reshape(-1, 4)[2:2 + circuit.len_z]
        z_start = [sum(int(a[k]) << (64 * k) for k in range(4)) for a in z0]
        chained = prev_end is None or prev_end == z_start
        print(f"{p}: {steps} steps, verify code {code}, starts where the previous proof ends: {chained}")
        ok = ok and code == 0 and chained
        prev_end, total = z_end, total + steps
    print("ACCEPTED" if ok else "REJECTED", f"({total} steps, final state {[hex(z) for z in prev_end]})")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
