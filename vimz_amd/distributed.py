"""Row-segment sharding of one fold over the GPUs of a node (BASELINE.json north_star: "independent row-folds shard
embarrassingly across the 8 GPUs ... host-side sequential final fold; no RCCL collectives needed").

One process per GPU.  Every rank folds a contiguous segment of rows into its own running instance; the only exchange
is the gather of the G exported instances on rank 0, which merges them in row order (relaxed+relaxed NIFS) and verifies.
The prover object is duck-typed (reset / fold / state_chain / export / merge / verify) so the same driver runs over the
GPU prover (vimz_amd.hip.Prover) and, in the CPU tests, over an oracle-backed stand-in.
"""
import numpy as np


def segment_bounds(n_steps, world):
    """Contiguous row segments, sizes differing by at most one (earlier ranks take the remainder)."""
    base, rem = divmod(n_steps, world)
    bounds, start = [], 0
    for r in range(world):
        size = base + (1 if r < rem else 0)
        bounds.append((start, start + size))
        start += size
    return bounds


def fold_sharded(prover, step_inputs, z0, rank=0, world=1, dist=None, verify=True):
    """Fold `step_inputs` (n, n_priv, 4) starting from IVC state z0 across `world` ranks.
    Returns on rank 0: dict(steps, z_final, verified); on other ranks: None."""
    n = len(step_inputs)
    lo, hi = segment_bounds(n, world)[rank]
    # state at which this rank's segment starts: hash-only chain over the rows before it (cheap; no folding)
    if lo > 0:
        zs = prover.state_chain(z0, step_inputs[:lo])
        z_start = [int(a[0]) | int(a[1]) << 64 | int(a[2]) << 128 | int(a[3]) << 192 for a in np.asarray(zs)[-1]]
    else:
        z_start = list(z0)
    prover.reset(z_start)
    if hi > lo:
        prover.fold(step_inputs[lo:hi])
    if world == 1:
        return {"steps": hi - lo, "verified": (prover.verify() == 0) if verify else None}
    blob = prover.export()
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(np.asarray(blob).tobytes(), gathered, dst=0)
    if rank != 0:
        return None
    for r in range(1, world):          # host-side sequential final fold, in row order
        prover.merge(np.frombuffer(gathered[r], dtype=np.uint8))
    return {"steps": n, "verified": (prover.verify() == 0) if verify else None}


def fold_local_segments(provers, step_inputs, z0, merge=True):
    """One GPU, several row segments folded CONCURRENTLY: a single sequential chain leaves most of an MI355X idle (the MSM
    tail is a latency chain), so a proof is split into len(provers) contiguous segments, each folded by its own prover
    (own context = own streams) from its own host thread (ctypes releases the GIL), then merged in row order on the device
    (vimz_prover_merge_prover).  Returns the merged prover (provers[0])."""
    from concurrent.futures import ThreadPoolExecutor
    n, S = len(step_inputs), len(provers)
    bounds = segment_bounds(n, S)
    starts = [list(z0)]
    for lo, hi in bounds[:-1]:
        zs = provers[0].state_chain(starts[-1], step_inputs[lo:hi])
        starts.append([int(a[0]) | int(a[1]) << 64 | int(a[2]) << 128 | int(a[3]) << 192 for a in np.asarray(zs)[-1]])
    for p, z in zip(provers, starts):
        p.reset(z)

    def work(i):
        lo, hi = bounds[i]
        if hi > lo:
            provers[i].fold(step_inputs[lo:hi])

    if S == 1:
        work(0)
    else:
        with ThreadPoolExecutor(S) as ex:
            list(ex.map(work, range(S)))
    if merge:
        for i in range(1, S):
            provers[0].merge_prover(provers[i])
    return provers[0]


def ivc_segments(ivcs, step_inputs, z0):
    """Split `step_inputs` into len(ivcs) contiguous row segments and return [(ivc, rows, z_start)], where z_start is the IVC
    state at which the segment begins (hash-only chain over the rows before it).  An image proven on several streams / GPUs is a
    list of IVC proofs, one per row segment, whose boundary states chain (z_end of segment j = z_start of segment j+1); they become
    ONE proof object by vimz_amd.hip.MergedProof.of(ivcs) (vimz_ivc_merge: the host-side sequential final fold)."""
    n, S = len(step_inputs), len(ivcs)
    bounds = segment_bounds(n, S)
    starts = [list(z0)]
    for lo, hi in bounds[:-1]:
        zs = ivcs[0].state_chain(starts[-1], step_inputs[lo:hi])
        starts.append([int(a[0]) | int(a[1]) << 64 | int(a[2]) << 128 | int(a[3]) << 192 for a in np.asarray(zs)[-1]])
    return [(ivcs[i], step_inputs[bounds[i][0]:bounds[i][1]], starts[i]) for i in range(S)]


def _ints(limbs):
    return [int(a[0]) | int(a[1]) << 64 | int(a[2]) << 128 | int(a[3]) << 192 for a in np.asarray(limbs)]


def fold_segments_merged(ivcs, step_inputs, z0, timings=None, merged_cls=None):
    """ONE proof of all rows from state z0 on one GPU: len(ivcs) contiguous row segments, each folded as a Nova IVC by its own prover
    (own context = own streams, own host thread), then merged in row order (vimz_ivc_merge: out-of-circuit NIFS on both curves).
    Segment j starts at the state segment j-1 ends in; that state comes from the hash-only chain over segment j-1's rows: the row
    digests of all but the last segment are computed at once, each by its successor's prover on ITS context, while segment 0
    already folds; only the short host chains over them are serial (a staggered start, not a serial prologue).  Returns the MergedProof (its verifier key is ivcs[0]: keep the IVCs open while it is in use).
    timings (optional dict): state_chain_s (host view, summed), merge_s."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    if merged_cls is None:
        from .hip import MergedProof as merged_cls
        if len(step_inputs) and all(hasattr(v, "h") for v in ivcs):       # the library's own fold_input: the same sequence in one C call
            merged, t = merged_cls.fold_segments(ivcs, step_inputs, z0)
            if timings is not None:
                timings["state_chain_s"] = timings.get("state_chain_s", 0.0) + t["state_chain_s"]
                timings["merge_s"] = timings.get("merge_s", 0.0) + t["merge_s"]
            return merged
    MergedProof = merged_cls
    n = len(step_inputs)
    used = [j for j, (lo, hi) in enumerate(segment_bounds(n, len(ivcs))) if hi > lo]
    bounds = segment_bounds(n, len(used))
    t_chain = 0.0
    z = [int(x) for x in z0]
    two_part = len(used) > 1 and all(hasattr(ivcs[j], "digest_stride") for j in used) and ivcs[used[0]].digest_stride() > 0
    with ThreadPoolExecutor(2 * max(1, len(used))) as ex:
        futs, dig = [], {}
        if two_part:      # the row hashes of segments 0 .. S-2, each on its successor's context, all at once: only the chains are serial
            for k in range(len(used) - 1):
                lo, hi = bounds[k]
                dig[k] = ex.submit(ivcs[used[k + 1]].row_digests, step_inputs[lo:hi])
        for k, j in enumerate(used):
            lo, hi = bounds[k]
            if k > 0:
                t0 = time.time()
                plo, phi = bounds[k - 1]
                if two_part:
                    z = _ints(ivcs[j].chain_from_digests(z, step_inputs[plo:phi], dig[k - 1].result())[-1])
                else:
                    z = _ints(ivcs[j].state_chain(z, step_inputs[plo:phi])[-1])
                t_chain += time.time() - t0
            ivcs[j].reset(z)
            futs.append(ex.submit(ivcs[j].fold, step_inputs[lo:hi]))
        for f in futs:
            f.result()
    # (merging segment k while later segments still fold measured worse: the merge's allocations and its large MSM on a
    #  high-priority stream stall the chains that are still running — 575-680 against 780 steps/s in a 20-row window)
    t0 = time.time()
    merged = MergedProof(ivcs[used[0]])
    try:
        for j in used[1:]:
            merged.merge(ivcs[j])
    except Exception:
        merged.close()
        raise
    t_merge = time.time() - t0
    if timings is not None:
        timings["state_chain_s"] = timings.get("state_chain_s", 0.0) + t_chain
        timings["merge_s"] = timings.get("merge_s", 0.0) + t_merge
    return merged


def prove_sharded(ivcs, step_inputs, z0, rank=0, world=1, dist=None, timings=None, merged_cls=None, shm_prefix=None):
    """ONE proof object of all rows from z0 over `world` ranks (one process per GPU; BASELINE.json north_star: "independent row-folds
    shard embarrassingly across the 8 GPUs ... host-side sequential final fold; no RCCL collectives needed").  Rank r proves the r-th
    contiguous run of rows with fold_segments_merged (len(ivcs) concurrent segments on its GPU); the runs' start states: every rank
    hashes its own rows, the digests are all-gathered, each rank chains over the rows before it (circuits whose digests depend on
    the state: one chain on rank 0, scattered); the ranks' merged proofs travel as bytes (gloo gather, or node-local shared memory when
    shm_prefix is given) and rank 0 folds them in row order (vimz_ivc_merge_merged).  Returns the proof on rank 0, None elsewhere.
    timings: state_chain_s (rank 0's chain + scatter, plus the local segments' chains), merge_s, final_fold_s."""
    import os
    import time
    if merged_cls is None:
        from .hip import MergedProof as merged_cls
    n = len(step_inputs)
    bounds = segment_bounds(n, world)
    z_start = [int(x) for x in z0]
    if world > 1:
        t0 = time.time()
        stride = ivcs[0].digest_stride() if hasattr(ivcs[0], "digest_stride") else 0
        if stride:
            # the chain in its two parts: every rank hashes ITS rows (the expensive, state-independent part: one GPU pass, side by
            # side on all GPUs), the digests are exchanged (a few hundred KB), and rank r runs the serial part — two or three small
            # permutations per row on the host — over the rows of the ranks before it.  (Rank 0 hashing everybody's rows on its GPU
            # cost 138 µs per row at 8K: 260 ms before the last of eight ranks could start a 650 ms fold.)
            lo, hi = bounds[rank]
            mine = np.ascontiguousarray(ivcs[0].row_digests(step_inputs[lo:hi])) if hi > lo else np.zeros((0, stride, 4), dtype=np.uint64)
            allg = [None] * world
            dist.all_gather_object(allg, mine.tobytes())
            for r in range(rank):
                plo, phi = bounds[r]
                if phi > plo:
                    dg = np.frombuffer(allg[r], dtype=np.uint64).reshape(phi - plo, stride, 4)
                    z_start = _ints(ivcs[0].chain_from_digests(z_start, step_inputs[plo:phi], dg)[-1])
        else:
            starts = [None] * world
            if rank == 0:
                z = starts[0] = z_start
                for r in range(1, world):
                    lo, hi = bounds[r - 1]
                    if hi > lo:
                        z = _ints(ivcs[0].state_chain(z, step_inputs[lo:hi])[-1])
                    starts[r] = z
            out = [None]
            dist.scatter_object_list(out, starts if rank == 0 else None, src=0)
            z_start = out[0]
        if timings is not None:
            timings["state_chain_s"] = timings.get("state_chain_s", 0.0) + time.time() - t0
    lo, hi = bounds[rank]
    proof = fold_segments_merged(ivcs, step_inputs[lo:hi], z_start, timings, merged_cls) if hi > lo else None
    if world == 1:
        return proof
    dist.barrier()                      # (every rank has its proof: what follows is the final fold alone, not the wait for the slowest rank)
    t0 = time.time()
    blob = np.asarray(proof.save()) if (proof is not None and rank > 0) else None
    if shm_prefix:
        if blob is not None:
            blob.tofile(f"{shm_prefix}{rank}")
        dist.barrier()
        blobs = [None] + [np.fromfile(f"{shm_prefix}{r}", dtype=np.uint8) if os.path.exists(f"{shm_prefix}{r}") else None for r in range(1, world)] if rank == 0 else None
    else:
        blobs = [None] * world if rank == 0 else None
        dist.gather_object(blob.tobytes() if blob is not None else None, blobs, dst=0)
    if rank != 0:
        if proof is not None:
            proof.close()
        if shm_prefix:
            dist.barrier()              # (rank 0 has read the files)
            if blob is not None:
                os.unlink(f"{shm_prefix}{rank}")
        return None
    for r in range(1, world):           # host-side sequential final fold, in row order
        if blobs[r] is None:
            continue
        other = merged_cls.load(ivcs[0], np.frombuffer(blobs[r], dtype=np.uint8) if isinstance(blobs[r], (bytes, bytearray)) else blobs[r])
        if proof is None:
            proof = other
        else:
            proof.merge(other)
            other.close()
    if shm_prefix:
        dist.barrier()
    if timings is not None:
        timings["final_fold_s"] = timings.get("final_fold_s", 0.0) + time.time() - t0
    return proof


def fold_concurrently(jobs):
    """jobs: [(ivc, rows)] folded from one host thread each (ctypes releases the GIL; every IVC has its own context = streams).
    A single IVC alternates between host work (the verifier circuits' witnesses) and GPU work; several of them interleave."""
    from concurrent.futures import ThreadPoolExecutor

    def work(j):
        ivc, rows = j
        if len(rows):
            ivc.fold(rows)

    if len(jobs) == 1:
        work(jobs[0])
    else:
        with ThreadPoolExecutor(len(jobs)) as ex:
            list(ex.map(work, jobs))
