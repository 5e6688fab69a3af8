#!/usr/bin/env python3
"""Verify a serialised proof in a process of its own (the `verify_folded_proof(proof, params, …)` side of the reference,
vimz/src/nova_snark_backend/folding.rs:45-56): rebuilds the public parameters (step circuit, both commitment keys) from the
transformation name and resolution, loads the merged proof (vimz_ivc_merged_load: one object for all row segments of the image) and
runs vimz_ivc_merged_verify.  The statement verified is the reference's and the VERIFIER's: iteration_count(transformation,
resolution) steps starting from Transformation::ivc_initial_state (factor / info given on the command line) — never the blob's.
usage: verify_proof.py <transformation> <resolution> <factor-or-info> <proof.merged.bin | proof.cfmerged.bin>
(write proofs with: tools/e2e.py <transformation> <resolution> <segments> ivc|cyclefold <prefix>  ->  <prefix>.merged.bin / <prefix>.cfmerged.bin;
 a merged Nova + CycleFold proof — the Sonobe backend's scheme — is recognised by its magic and verified by vimz_cf_merged_verify)"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from vimz_amd import _lib, folding, hip  # noqa: E402,F401


def main():
    t, res, extra, path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    z_first = folding.ivc_initial_state(t, {"factor": extra, "info": extra})
    want_steps = folding.iteration_count(t, res)
    ctx = hip.Context(0)
    blob = np.fromfile(path, dtype=np.uint8)
    cyclefold = int.from_bytes(blob[:8].tobytes(), "little") == 0x32474d46435a56      # "VZCFMG2": a merged Nova + CycleFold proof (vimz_cf_merged_save)
    circuit, params = folding.prepare_folding(ctx, t, res, backend="sonobe" if cyclefold else "nova-snark")
    if cyclefold:
        vk = hip.CycleFoldIVC(ctx, circuit, params.ck, params.secondary_key(), max_batch=1)
        proof = hip.CycleFoldMerged.load(vk, blob)
    else:
        vk = hip.IVC(ctx, circuit, params.ck, params.secondary_key(), max_batch=1)
        proof = hip.MergedProof.load(vk, blob)
    code = proof.verify(want_steps, z_first)
    zs, ze, n = proof.state()
    print(f"{path}: {proof.info()['segments']} segments, {n} steps, verify code {code} for ({want_steps} steps from the transformation's initial state)")
    print("ACCEPTED" if code == 0 else "REJECTED", f"(final state {[hex(z) for z in ze]})")
    sys.exit(0 if code == 0 else 1)


if __name__ == "__main__":
    main()
