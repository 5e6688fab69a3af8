#!/usr/bin/env python3
"""End-to-end run of one transformation on the GPU, the way `vimz -b nova-snark -f <t>` sequences it
(vimz/src/nova_snark_backend/mod.rs:22-80): prepare input -> prepare folding (circuit + key) -> fold every row -> verify.
Prints the span times the reference logs ("Prepare input", "Prepare folding", "Fold input", "Verify folded proof").
usage: e2e.py <transformation> <resolution> [segments] [ivc|accumulator|cyclefold] [proof file prefix or -] [witness batch]
ivc (default): ONE proof object — the rows are proven as `segments` Nova IVCs of contiguous row segments folded concurrently and
merged (vimz_ivc_merge), then compressed;
accumulator: NIFS accumulators of the segments merged by a final fold;
cyclefold: the Sonobe backend's sequence (vimz/src/sonobe_backend/mod.rs:52-95: prepare folding, fold input, verify folded proof) with Nova +
CycleFold; with one segment also the decider (Prepare decider, Generate decider proof: vimz_decider_*; the full one, or VIMZ_E2E_DECIDER=light) and the calldata bytes."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402
from vimz_amd import _lib, folding, hip  # noqa: E402
from vimz_amd.distributed import fold_local_segments, fold_segments_merged  # noqa: E402


def main():
    t, res = sys.argv[1], sys.argv[2]
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 3      # (a lone, cold image: three segments — a fourth pays for itself only over repeated proofs, profiles/r06_segments_sweep.txt)
    mode = sys.argv[4] if len(sys.argv) > 4 else "ivc"
    save = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] != "-" else None
    batch = int(sys.argv[6]) if len(sys.argv) > 6 else 0            # rows whose witnesses are generated together (0: folding.default_batch)
    spans = {}
    t_wall = time.time()
    # "Prepare input" (image -> packed rows, host) runs on a thread of its own under "Prepare folding" (contexts, circuit, keys, provers): neither needs the
    # other.  The spans are each phase's own duration; `wall_total_s` is the run's real length.
    import threading
    inp = {}

    def _prepare_input():
        t_i = time.time()
        inp["rows"], inp["z0"] = bench.build_inputs(t, res)
        inp["s"] = time.time() - t_i
    th_in = threading.Thread(target=_prepare_input)
    th_in.start()
    t0 = time.time()
    # (contexts, step circuit, keys and the segments' provers side by side: folding.prepare_folding_overlapped)
    pmode = mode if mode in ("ivc", "cyclefold") else "accumulator"
    ctxs, circuit, params, made, setup_split = folding.prepare_folding_overlapped(0, S, t, res, mode=pmode, batch=batch)
    batch = batch or folding.default_batch(circuit)
    th_in.join()
    rows, z0 = inp["rows"], inp["z0"]
    spans["Prepare input"] = inp["s"]
    if mode == "ivc":
        ck2 = params.secondary_key()
        ivcs = made
        spans["Prepare folding"] = time.time() - t0
        t0 = time.time()
        tm = {}
        proof = fold_segments_merged(ivcs, rows, z0, tm)      # ONE proof object: S concurrent segments + merge (S = 1: one chain)
        for c in ctxs:
            c.sync()
        spans["Fold input"] = time.time() - t0
        t0 = time.time()
        ok = proof.verify(len(rows), z0) == 0
        spans["Verify folded proof"] = time.time() - t0
        zs, ze, n = proof.state()
        # the reference's remaining spans (vimz/src/nova_snark_backend/mod.rs:52-67): RecursiveSNARK -> CompressedSNARK
        blob, tc = proof.compress()
        spans["Prepare compression"] = tc["setup_s"]
        spans["Compress proof"] = tc["prove_s"]
        t0 = time.time()
        ok = ok and hip.MergedProof.verify_compressed(ivcs[0], blob, len(rows), z0) == 0
        spans["Verify compressed proof"] = time.time() - t0
        spans["compressed proof bytes"] = int(len(blob))
        if save:
            proof.save().tofile(f"{save}.merged.bin")       # verify elsewhere: tools/verify_proof.py
        print(json.dumps({"config": f"{t}_step_{res}", "mode": "ivc", "steps": n, "segments": S, "witness_batch": batch, "proof_objects": 1, "verified": ok, "spans_s": spans, "prepare_folding_split_s": setup_split,
                          "state_chain_s": tm.get("state_chain_s"), "merge_s": tm.get("merge_s"),
                          "steps_per_s": n / spans["Fold input"], "total_s": sum(v for k, v in spans.items() if not k.endswith("bytes")), "wall_total_s": time.time() - t_wall,
                          "final_state": [hex(z) for z in ze]}))
        return
    if mode == "cyclefold":
        ck2 = params.secondary_key()
        cfs = made
        spans["Prepare folding"] = time.time() - t0
        # Decider::preprocess (mod.rs:72-75) depends on the shapes only, not on the fold: its host part — circuit synthesis, the QAP at the trapdoor,
        # 0.6 s at contrast HD — runs on a thread of its own UNDER the fold; its GPU part (the key's points) takes the context when the fold has left it
        dec_box, dec_thread = {}, None
        import os
        light_decider = os.environ.get("VIMZ_E2E_DECIDER", "full") == "light"      # (the reference's opt-in `light-test` feature, vimz/Cargo.toml:56-59)
        if S == 1:
            import threading

            def _prep_decider():
                t_d = time.time()
                try:
                    dec_box["dec"] = hip.Decider(cfs[0], kzg_vk=params.kzg_vk, light=light_decider)
                except BaseException as e:      # noqa: BLE001
                    dec_box["err"] = e
                dec_box["seconds"] = time.time() - t_d
            dec_thread = threading.Thread(target=_prep_decider)
            dec_thread.start()
        t0 = time.time()
        rows_a = np.stack(rows)
        tm = {}
        if S > 1:      # segments' start states staggered under the folds (row digests of all but the last segment at once), then the merge
            proof = fold_segments_merged(cfs, rows_a, z0, tm, merged_cls=hip.CycleFoldMerged)
        else:
            cfs[0].reset(z0); cfs[0].fold(rows_a)
            proof = cfs[0]
        t_chain, t_merge = tm.get("state_chain_s", 0.0), tm.get("merge_s", 0.0)
        spans["Fold input"] = time.time() - t0
        t0 = time.time()
        ok = proof.verify(len(rows), z0) == 0
        spans["Verify folded proof"] = time.time() - t0
        ze = proof.state()[1] if S > 1 else proof.state()[0]
        if save and S > 1:
            proof.save().tofile(f"{save}.cfmerged.bin")       # verify elsewhere: tools/verify_proof.py
        decider = None
        if S == 1:      # the reference's next spans (mod.rs:72-80): Decider::preprocess, Decider::prove, verify_final_proof -> the calldata (solidity.rs:13-27)
            from vimz_amd import calldata
            t0 = time.time()
            dec_thread.join()
            if "err" in dec_box:
                raise dec_box["err"]
            dec = dec_box["dec"]
            spans["Prepare decider"] = time.time() - t0      # (what is left of it after the fold and its verification; the thread's own time: decider["prepare_thread_s"])
            t0 = time.time()
            raw, dd = calldata.decider_calldata(dec)
            spans["Generate decider proof"] = time.time() - t0
            t0 = time.time()
            dec_ok = dec.verify(dd["steps"], dd["z0"], dd["z_i"], dd["words"])        # verify_final_proof (decider.rs:31-50): the contract's checks, locally
            spans["Verify decider proof"] = time.time() - t0
            if dec_ok != 0:
                raise SystemExit(f"the decider proof does not verify: result bits {dec_ok}")
            decider = {"variant": "light" if light_decider else "full", "circuit": dec.info(), "setup_s": dec.setup_seconds, "prepare_thread_s": dec_box["seconds"], "prove_s": dd["seconds"], "calldata_bytes": len(raw), "verified": dec_ok == 0,
                       "note": "Groth16 over BN254 for this library's decider circuit (contract's public-input layout; locally trusted setup), final fold + KZG openings on the GPU; "
                               "verified by vimz_decider_verify = the checks of contracts/*Verifier.sol (tests/_novadecider.py restates the contract and is pinned on the reference's six proofs)"}
            if save:
                open(f"{save}.calldata.bin", "wb").write(raw)
            dec.close()
        print(json.dumps({"config": f"{t}_step_{res}", "mode": "cyclefold", "steps": len(rows), "segments": S, "witness_batch": batch, "proof_objects": 1, "verified": ok, "spans_s": spans, "prepare_folding_split_s": setup_split,
                          "state_chain_s": t_chain, "merge_s": t_merge, "info": cfs[0].info(), "decider": decider,
                          "ms_per_step_first_segment": {k: 1e3 * sec / max(1, cfs[0].info()["steps"]) for k, (sec, n) in cfs[0].profile().items()},
                          "steps_per_s": len(rows) / spans["Fold input"], "total_s": sum(spans.values()), "wall_total_s": time.time() - t_wall, "final_state": [hex(z) for z in ze]}))
        return
    provers = made
    spans["Prepare folding"] = time.time() - t0
    t0 = time.time()
    merged = fold_local_segments(provers, rows, z0)
    for c in ctxs:
        c.sync()
    spans["Fold input"] = time.time() - t0
    t0 = time.time()
    ok = merged.verify() == 0
    spans["Verify folded proof"] = time.time() - t0
    inst = merged.instance()
    print(json.dumps({"config": f"{t}_step_{res}", "mode": "accumulator", "steps": inst["steps"], "segments": S, "verified": ok, "spans_s": spans,
                      "steps_per_s": inst["steps"] / spans["Fold input"], "total_s": sum(spans.values()), "wall_total_s": time.time() - t_wall,
                      "final_state": [hex(int(a[0]) | int(a[1]) << 64 | int(a[2]) << 128 | int(a[3]) << 192) for a in inst["z"]]}))


if __name__ == "__main__":
    main()
