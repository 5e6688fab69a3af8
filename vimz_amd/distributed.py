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
