"""Host-side mirror of the reference's nova-snark backend glue (vimz/src/nova_snark_backend/{input,folding}.rs):
same function names, argument meaning and error behaviour, driving the GPU through the C ABI.

    prepare_input   (input.rs:25-40)    VIMzInput -> per-step inputs + z0
    prepare_folding (folding.rs:20-25)  load_r1cs + create_public_params -> (circuit, FoldingParams)
    fold_input      (folding.rs:27-43)  create_recursive_circuit -> FoldingProof
    verify_folded_proof (folding.rs:45-56)
    compress_proof / verify_compressed_proof (mod.rs:52-67)  CompressedSNARK::{setup, prove, verify}
"""
import time

import numpy as np

from . import _lib
from .circuit import Circuit, default_shape

DEMO_STEPS = 10  # vimz/src/lib.rs:9
AUGMENTED_ROOM = 8192   # wires / constraints Nova's verifier circuit adds to a step circuit (vimz_ivc_create)
CYCLEFOLD_ROOM = 28672  # ... and Nova + CycleFold's main circuit F' (vimz_cf_create: 27.5 k — non-native folds of two CycleFold instances per step)
SECONDARY_KEY_LEN = 1 << 13   # generators of the secondary (Grumpkin) key: the secondary circuit has 7.6 k wires / rows

# vimz/src/transformation.rs:93-123
ITERATION_COUNT = {"SD": 480, "HD": 720, "FHD": 1080, "4K": 2160, "8K": 4320}
ITERATION_COUNT_BLOCK = {"HD": 576}
RATIO_TO_LOWER = {"HD": (3, 2), "FHD": (3, 2), "4K": (2, 1), "8K": (2, 1)}


def ivc_initial_state(transformation, inp):
    """Transformation::ivc_initial_state (vimz/src/transformation.rs:25-40)."""
    if transformation in ("blur", "sharpness"):
        return [0, 0, 0, 0]
    if transformation in ("brightness", "contrast"):
        return [0, 0, int(inp["factor"])]
    if transformation == "crop":
        return [0, 0, int(inp["info"])]
    if transformation in ("grayscale", "redact", "resize"):
        return [0, 0]
    if transformation == "hash":
        return [0]
    raise ValueError(transformation)


def iteration_count(transformation, resolution):
    if transformation == "redact":
        return ITERATION_COUNT_BLOCK[resolution]
    if transformation == "resize":
        return ITERATION_COUNT[resolution] // RATIO_TO_LOWER[resolution][0]
    return ITERATION_COUNT[resolution]


def prepare_step_input(step, transformation, inp, resolution):
    """prepare_step_input (input.rs:57-96): flattened private inputs of one step as (n, 4) limbs."""
    o, t = inp["original"], inp.get("transformed")
    if transformation in ("brightness", "contrast", "grayscale"):
        return np.concatenate([o[step], t[step]])
    if transformation in ("blur", "sharpness"):
        return np.concatenate([o[step:step + 3].reshape(-1, 4), t[step]])
    if transformation in ("crop", "hash"):
        return o[step]
    if transformation == "redact":
        ind = np.array([[int(inp["redact"][step]), 0, 0, 0]], dtype=np.uint64)
        return np.concatenate([o[step], ind])
    if transformation == "resize":
        a, b = RATIO_TO_LOWER[resolution]
        return np.concatenate([o[step * a:(step + 1) * a].reshape(-1, 4), t[step * b:(step + 1) * b].reshape(-1, 4)])
    raise ValueError(transformation)


def prepare_input(transformation, inp, resolution="HD", demo=False):
    """prepare_input (input.rs:25-40): returns (ivc_step_inputs (steps, n_priv, 4), initial_state)."""
    n = iteration_count(transformation, resolution)
    if demo:
        n = min(n, DEMO_STEPS)
    steps = np.stack([prepare_step_input(i, transformation, inp, resolution) for i in range(n)])
    return steps, ivc_initial_state(transformation, inp)


class FoldingParams:
    """The analogue of nova-snark's PublicParams: R1CS shape + both commitment keys (BN254 G1 for the primary circuit, Grumpkin for
    the secondary), resident on one GPU."""

    def __init__(self, ctx, circuit, ck, keygen_seconds, ck_secondary=None, kzg_vk=None):
        self.ctx, self.circuit, self.ck, self.keygen_seconds, self.ck_secondary = ctx, circuit, ck, keygen_seconds, ck_secondary
        self.kzg_vk = kzg_vk      # backend "sonobe": [tau]G2 of the KZG SRS `ck` is (the decider's verifier needs it)

    def secondary_key(self):
        if self.ck_secondary is None:
            self.ck_secondary = self.ctx.bases_generate(_lib.CURVE_GRUMPKIN, SECONDARY_KEY_LEN, b"ck-secondary")
        return self.ck_secondary

    def free(self):
        if self.ck_secondary is not None:
            self.ck_secondary.free()
            self.ck_secondary = None
        self.ck.free()


def prepare_folding(ctx, transformation, resolution="HD", ck_label=b"ck", window_tables=15, backend="nova-snark"):
    """prepare_folding (folding.rs:20-25): build the step circuit and derive the commitment key on the GPU.
    window_tables: 13..16 = window tables T_j[i] = 2^(c j)·P_i of the key in HBM with ONE bucket set shared by all windows
    (vimz_bases_precompute; default 15: 17 x the key = 0.7 GB at HD, 17 digits per scalar instead of 24 — the large MSM of a step
    1.10 -> 0.84 ms alone, the three-segment bench 964 -> 1095 steps/s, profiles/r04_msm_phases_tables.txt); 11 = tables with the usual
    per-window buckets (no Horner on the host); 0 = none.
    backend: "nova-snark" (vimz/src/nova_snark_backend) or "sonobe" (vimz/src/sonobe_backend: Nova + CycleFold, fold_input(mode="cyclefold"));
    the reference's two backends size their keys the same way (for the augmented circuit); Sonobe's is a KZG SRS (`KZG::setup` inside
    `prepare_folding`, vimz/src/sonobe_backend/folding.rs:36-48): powers of a tau drawn from the OS's randomness and forgotten
    (vimz_kzg_setup), whose [tau]G2 is kept in FoldingParams.kzg_vk for the decider's verifier."""
    t0 = time.time()
    circuit = Circuit(transformation, *default_shape(transformation, resolution))
    # next power of two, as nova-snark sizes ck — of the AUGMENTED circuit: the verifier circuit adds 7.7 k wires / rows
    n = 1 << (max(circuit.n_wires, circuit.n_constraints) + (CYCLEFOLD_ROOM if backend == "sonobe" else AUGMENTED_ROOM) - 1).bit_length()
    kzg_vk = None
    if backend == "sonobe":
        from . import hip
        ck, kzg_vk = hip.kzg_setup(ctx, n)
    else:
        ck = ctx.bases_generate(_lib.CURVE_BN254_G1, n, ck_label)
    if window_tables:
        ck.precompute(16 if window_tables is True else int(window_tables))
    return circuit, FoldingParams(ctx, circuit, ck, time.time() - t0, kzg_vk=kzg_vk)


def prepare_folding_overlapped(device, segments, transformation, resolution="HD", window_tables=15, backend="nova-snark", mode="ivc", batch=0, ctxs=None):
    """prepare_folding for a proof made as `segments` concurrent row segments, with its parts side by side instead of one after another
    (once the fold takes 0.6 s the set-up is the largest span of an HD run).  What depends on what:
        contexts (HIP runtime, streams)      — nothing; one thread each
        step circuit (host builder, 0.09 s)   — nothing: runs under the contexts' creation
        commitment key + window tables (GPU) — the circuit's size, context 0
        one prover per segment                — circuit, keys, its context: the verifier circuits' synthesis (host) and the device buffers of the
                                                segments' provers are made side by side (ctypes releases the GIL around every library call)
    Returns (ctxs, circuit, params, provers, seconds: {contexts_and_circuit, keys, provers, total}).  mode: "ivc" (hip.IVC), "cyclefold"
    (hip.CycleFoldIVC over a KZG SRS), "accumulator" (hip.Prover).  Sequential equivalent: prepare_folding + one prover per context."""
    import threading
    from . import hip
    t_all = time.time()
    out, err = {}, []

    def guarded(fn):
        def run(*a):
            try:
                fn(*a)
            except BaseException as e:      # noqa: BLE001 (re-raised on the calling thread)
                err.append(e)
        return run

    own_ctxs = ctxs is None
    ctxs = [None] * segments if own_ctxs else list(ctxs)

    def make_ctx(k):
        ctxs[k] = hip.Context(device)

    circuit_built = threading.Event()

    def make_circuit():
        try:
            out["circuit"] = Circuit(transformation, *default_shape(transformation, resolution))
        finally:
            circuit_built.set()      # (the keys need its SIZE only: they are made while this thread goes on)
        if mode == "ivc":      # (the augmented circuits + their digests: 0.12 s of host work the segments' IVCs share — under the keys' generation)
            out["circuit"].prepare_ivc()

    # the step circuit and contexts 1.. on threads; context 0 here: the keys need only it and the circuit's size, and are made while the others still come up
    th_circ = threading.Thread(target=guarded(make_circuit))
    th_ctx = [threading.Thread(target=guarded(make_ctx), args=(k,)) for k in range(1, segments)] if own_ctxs else []
    for x in [th_circ] + th_ctx:
        x.start()
    if own_ctxs:
        guarded(make_ctx)(0)
    circuit_built.wait()
    if err or "circuit" not in out:
        th_circ.join()
        for x in th_ctx:
            x.join()
        raise err[0] if err else RuntimeError("the step circuit could not be built")
    circuit = out["circuit"]
    t_cc = time.time()
    sonobe = backend == "sonobe" or mode == "cyclefold"
    n = 1 << (max(circuit.n_wires, circuit.n_constraints) + (CYCLEFOLD_ROOM if sonobe else AUGMENTED_ROOM) - 1).bit_length()
    kzg_vk = None
    if sonobe:
        ck, kzg_vk = hip.kzg_setup(ctxs[0], n)
    else:
        ck = ctxs[0].bases_generate(_lib.CURVE_BN254_G1, n, b"ck")
    if window_tables:
        ck.precompute(16 if window_tables is True else int(window_tables))
    for x in th_ctx + [th_circ]:
        x.join()
    if err:
        raise err[0]
    params = FoldingParams(ctxs[0], circuit, ck, time.time() - t_cc, kzg_vk=kzg_vk)
    ck2 = params.secondary_key() if mode in ("ivc", "cyclefold") else None      # (mode "none": keys only, the caller makes its provers)
    t_keys = time.time()
    batch = batch or default_batch(circuit)
    provers = [None] * segments

    def make_prover(k):
        if mode == "none":
            return
        if mode == "ivc":
            provers[k] = hip.IVC(ctxs[k], circuit, ck, ck2, max_batch=batch)
        elif mode == "cyclefold":
            provers[k] = hip.CycleFoldIVC(ctxs[k], circuit, ck, ck2, max_batch=batch)
        else:
            provers[k] = hip.Prover(ctxs[k], circuit, ck, max_batch=batch)

    th = [threading.Thread(target=guarded(make_prover), args=(k,)) for k in range(segments)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    if err:
        raise err[0]
    t_end = time.time()
    return ctxs, circuit, params, provers, {"contexts_and_circuit": t_cc - t_all, "keys": t_keys - t_cc, "provers": t_end - t_keys, "total": t_end - t_all}


def default_batch(circuit):
    """Rows whose witnesses are generated together (and whose (A,B,C)·z and commitments the producer keeps ahead of the folds).  A batch costs one
    Poseidon-chain latency whatever its size, so wide circuits want large batches — and a batch buffer is rows x wires x 32 B, three provers per GPU.
    Round 5, measured as a trade-off (profiles/r05_batch_sweep.txt; three provers, steps/s over 256 / 128 / 128 rows and peak device memory):
        contrast HD   32: 1 160 / 13.8 GB    64: 1 181 / 21.4 GB    128: 1 219 / 35.6 GB
        contrast 4K   32:   673 / 31.4 GB    64:   693 / 52.8 GB    128:   704 / 97.6 GB
        resize 8K     32:   583 / 29.6 GB    64:   613 / 49.2 GB    128:   610 / 90.3 GB
    64 keeps every BASELINE configuration below 64 GB at a loss of at most 1.6 % against 128 (which was the default above 500 k wires until round 4:
    98-105 GB for three provers).  `--batch` / max_batch override it."""
    return 64


class FoldingProof:
    """FoldingProof (folding.rs:17): the RecursiveSNARK (mode "ivc": a vimz_ivc) or, in accumulator mode, the NIFS accumulator of
    the step circuit's own instances (a vimz_prover; mergeable across row segments, not a RecursiveSNARK)."""

    def __init__(self, prover, steps, z0, mode):
        self.prover, self.steps, self.z0, self.mode = prover, steps, z0, mode

    def instance(self):
        return self.prover.instance()

    def state(self):
        """The final IVC state z_n as integers."""
        if self.mode in ("ivc", "cyclefold"):
            return self.prover.state()[0]
        if self.mode in ("merged", "cyclefold-merged"):
            return self.prover.state()[1]
        return _limbs_to_ints(self.prover.instance()["z"])

    def close(self):
        """Release the proof and, for a merged proof, the segment provers and contexts fold_input created for it."""
        self.prover.close()
        for o in getattr(self, "_owned", []):
            o.close()
        self._owned = []


def _limbs_to_ints(a):
    return [sum(int(x[k]) << (64 * k) for k in range(4)) for x in np.asarray(a).reshape(-1, 4)]


def fold_input(params, ivc_step_inputs, initial_state, max_batch=None, prover=None, mode="ivc", segments=1):
    """fold_input (folding.rs:27-43): one RecursiveSNARK over all steps (mode "ivc", what the reference produces), or the NIFS
    accumulator (mode "accumulator").  Raises VimzError (the reference panics with "Failed to fold input").
    segments = S > 1 (mode "ivc"): the rows are proven as S contiguous segments folded CONCURRENTLY on this GPU (one IVC each, own
    context and streams) and merged into ONE proof object (vimz_ivc_merge; FoldingProof.mode == "merged") — a single chain leaves a
    quarter of an MI355X idle."""
    from .hip import IVC, Context, CycleFoldIVC, MergedProof, Prover
    if max_batch is None:
        max_batch = default_batch(params.circuit)
    if mode == "cyclefold" and prover is None and segments > 1 and len(ivc_step_inputs) >= segments:
        # S contiguous row segments folded concurrently (one CycleFold prover each, own context and streams) and merged into ONE object
        from .distributed import fold_concurrently, ivc_segments
        from .hip import CycleFoldMerged
        ctxs = [params.ctx] + [Context(params.ctx.device) for _ in range(segments - 1)]
        ck2 = params.secondary_key()
        cfs = [CycleFoldIVC(c, params.circuit, params.ck, ck2, max_batch=max_batch) for c in ctxs]
        try:
            from .distributed import fold_segments_merged
            merged = fold_segments_merged(cfs, ivc_step_inputs, initial_state, None, merged_cls=CycleFoldMerged)      # (start states staggered under the folds)
        except Exception:
            for o in cfs + ctxs[1:]:
                o.close()
            raise
        proof = FoldingProof(merged, len(ivc_step_inputs), list(initial_state), "cyclefold-merged")
        proof._owned = cfs + ctxs[1:]
        proof.verifier_key = cfs[0]
        return proof
    if mode == "cyclefold":        # the Sonobe backend's fold_input (vimz/src/sonobe_backend/folding.rs:52-66): Nova + CycleFold prove_step per row
        p = prover if prover is not None else CycleFoldIVC(params.ctx, params.circuit, params.ck, params.secondary_key(), max_batch=max_batch)
        p.reset(initial_state)
        p.fold(ivc_step_inputs)
        return FoldingProof(p, len(ivc_step_inputs), list(initial_state), "cyclefold")
    if prover is None and mode == "ivc" and segments > 1 and len(ivc_step_inputs) >= segments:
        from .distributed import fold_concurrently, ivc_segments
        ctxs = [params.ctx] + [Context(params.ctx.device) for _ in range(segments - 1)]
        ck2 = params.secondary_key()
        ivcs = [IVC(c, params.circuit, params.ck, ck2, max_batch=max_batch) for c in ctxs]
        try:
            segs = ivc_segments(ivcs, ivc_step_inputs, initial_state)
            for v, rows, z in segs:
                v.reset(z)
            fold_concurrently([(v, rows) for v, rows, z in segs])
            merged = MergedProof.of(ivcs)
        except Exception:
            for o in ivcs + ctxs[1:]:
                o.close()
            raise
        proof = FoldingProof(merged, len(ivc_step_inputs), list(initial_state), "merged")
        proof._owned = ivcs + ctxs[1:]          # (the first IVC is the merged proof's verifier key: released after it)
        proof.verifier_key = ivcs[0]
        return proof
    if prover is not None:
        p = prover
        mode = "ivc" if isinstance(prover, IVC) else "accumulator"
    elif mode == "ivc":
        p = IVC(params.ctx, params.circuit, params.ck, params.secondary_key(), max_batch=max_batch)
    else:
        p = Prover(params.ctx, params.circuit, params.ck, max_batch=max_batch)
    p.reset(initial_state)
    p.fold(ivc_step_inputs)
    return FoldingProof(p, len(ivc_step_inputs), list(initial_state), mode)


def verify_folded_proof(proof, params, num_steps, initial_state):
    """verify_folded_proof (folding.rs:45-56: RecursiveSNARK::verify(pp, num_steps, z0, [0])); raises like the reference's
    expect("Failed to verify folded proof")."""
    if proof.mode in ("ivc", "merged", "cyclefold", "cyclefold-merged"):      # (cyclefold: verify_folding, vimz/src/sonobe_backend/folding.rs:69-75)
        r = proof.prover.verify(num_steps, initial_state)
        if r != 0:
            raise _lib.VimzError(_lib.ERR_UNSAT, f"Failed to verify folded proof (flags {r:#x})")
        return
    inst = proof.prover.instance()
    if inst["steps"] != num_steps or list(initial_state) != list(proof.z0):
        raise _lib.VimzError(_lib.ERR_INVALID, "Failed to verify folded proof: step count / initial state mismatch")
    r = proof.prover.verify()
    if r != 0:
        raise _lib.VimzError(_lib.ERR_UNSAT, f"Failed to verify folded proof (flags {r:#x})")


def compress_proof(params, proof):
    """CompressedSNARK::setup + prove (vimz/src/nova_snark_backend/mod.rs:52-59; spans "Prepare compression" / "Compress proof").
    Returns (proof bytes, timings)."""
    if proof.mode not in ("ivc", "merged"):
        raise _lib.VimzError(_lib.ERR_INVALID, "Failed to compress proof: only a RecursiveSNARK (mode \"ivc\", or merged segments) can be compressed")
    return proof.prover.compress()


def verify_compressed_proof(verifier_key, compressed, num_steps, initial_state):
    """compressed_proof.verify(&vk, num_steps, initial_state, secondary_initial_state) (mod.rs:63-67).  verifier_key: an IVC object
    created for the same step circuit and keys (e.g. proof.prover, or a fresh hip.IVC in another process)."""
    if isinstance(compressed, (bytes, bytearray, memoryview)):
        compressed = np.frombuffer(bytes(compressed), dtype=np.uint8)
    head = np.ascontiguousarray(compressed, dtype=np.uint8).reshape(-1)[:8]
    if head.size < 8:
        raise _lib.VimzError(_lib.ERR_INVALID, "Failed to verify proof: the blob is shorter than its header")
    words = np.frombuffer(head.tobytes(), dtype=np.uint64)
    if int(words[0]) == 0x31474D43565A:      # a compressed MERGED proof (vimz_ivc_merged_compress)
        from .hip import MergedProof
        r = MergedProof.verify_compressed(verifier_key, compressed, num_steps, initial_state)
    else:
        r = verifier_key.verify_compressed(compressed, num_steps, initial_state)
    if r != 0:
        raise _lib.VimzError(_lib.ERR_UNSAT, f"Failed to verify proof (flags {r:#x})")
