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


def fold_segments_merged(ivcs, step_inputs, z0, timings=None, merged_cls=None, digests=None):
    """ONE proof of all rows from state z0 on one GPU: len(ivcs) contiguous row segments, each folded as a Nova IVC by its own prover
    (own context = own streams, own host thread), then merged in row order (vimz_ivc_merge: out-of-circuit NIFS on both curves).
    Segment j starts at the state segment j-1 ends in; that state comes from the hash-only chain over segment j-1's rows: the row
    digests of all but the last segment are computed at once, each by its successor's prover on ITS context, while segment 0
    already folds; only the short host chains over them are serial (a staggered start, not a serial prologue).  Returns the MergedProof (its verifier key is ivcs[0]: keep the IVCs open while it is in use).
    timings (optional dict): state_chain_s (host view, summed), merge_s."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    if merged_cls is None:
        from . import hip as _hip
        merged_cls = {_hip.IVC: _hip.MergedProof, _hip.CycleFoldIVC: _hip.CycleFoldMerged}.get(type(ivcs[0])) if ivcs else None
        if merged_cls is None or any(type(v) is not type(ivcs[0]) for v in ivcs):
            raise TypeError("fold_segments_merged: give merged_cls for provers that are not all vimz_amd.hip.IVC or all vimz_amd.hip.CycleFoldIVC")
    from . import hip as _hip2
    if merged_cls is _hip2.MergedProof:
        if not all(isinstance(v, _hip2.IVC) for v in ivcs):
            raise TypeError("fold_segments_merged: MergedProof merges vimz_amd.hip.IVC provers only (CycleFold provers: merged_cls=CycleFoldMerged)")
        if len(step_inputs):       # the library's own fold_input: the same sequence in one C call
            merged, t = merged_cls.fold_segments(ivcs, step_inputs, z0, digests=digests)
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
                dig[k] = ex.submit((lambda a, b: digests[a:b]) if digests is not None else (lambda a, b, v=ivcs[used[k + 1]]: v.row_digests(step_inputs[a:b])), lo, hi)
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


def _send_bytes(dist, b, dst):
    """A small message to one rank (gloo point-to-point: length, then payload)."""
    import torch
    b = bytes(b)
    dist.send(torch.tensor([len(b)], dtype=torch.int64), dst)
    if b:
        dist.send(torch.frombuffer(bytearray(b), dtype=torch.uint8), dst)


def _recv_bytes(dist, src):
    import torch
    n = torch.zeros(1, dtype=torch.int64)
    dist.recv(n, src)
    if int(n[0]) == 0:
        return b""
    buf = torch.empty(int(n[0]), dtype=torch.uint8)
    dist.recv(buf, src)
    return buf.numpy().tobytes()


def tree_rounds(world):
    """The pairwise tree of the ranks' final fold: [(receiver, sender)] per round; in round k rank r with r % 2^(k+1) == 2^k hands its
    proof to rank r - 2^k.  Runs of rows stay adjacent at every node (receiver's rows end where the sender's begin); depth ceil(log2 world)."""
    rounds, step = [], 1
    while step < world:
        rounds.append([(r, r + step) for r in range(0, world, 2 * step) if r + step < world])
        step *= 2
    return rounds


def _offer(proof, merged_cls, transport, shm_dir):
    """What the handing-over rank sends: {"kind": "none" | "ipc" | "file" | "bytes", ...} (+ a binary payload) and the file it must remove afterwards."""
    import os
    import tempfile
    if proof is None:
        return {"kind": "none"}, b"", None
    if transport == "ipc":
        return {"kind": "ipc"}, np.asarray(proof.share()).tobytes(), None
    blob = np.asarray(proof.save())
    if transport == "file":
        fd, path = tempfile.mkstemp(prefix="vimz_proof_", dir=shm_dir)      # (O_EXCL, unpredictable name, this user only)
        with os.fdopen(fd, "wb") as f:
            blob.tofile(f)
        return {"kind": "file", "path": path, "bytes": int(blob.size)}, b"", path
    return {"kind": "bytes"}, blob.tobytes(), None


def _encode_offer(header, payload=b""):
    """A hand-over message: 4-byte little-endian header length ‖ JSON header ‖ binary payload (an IPC ticket or a proof blob).  No pickle: what a
    peer sends is parsed as data, never executed."""
    import json
    h = json.dumps(header).encode()
    return len(h).to_bytes(4, "little") + h + bytes(payload)


def _decode_offer(raw):
    import json
    raw = bytes(raw)
    if len(raw) < 4:
        raise ValueError("hand-over message: truncated")
    n = int.from_bytes(raw[:4], "little")
    if n > len(raw) - 4 or n > 1 << 16:
        raise ValueError("hand-over message: bad header length")
    header = json.loads(raw[4:4 + n].decode())
    if not isinstance(header, dict) or header.get("kind") not in ("none", "ipc", "file", "bytes", "abort"):
        raise ValueError("hand-over message: unknown kind")
    return header, raw[4 + n:]


def _abort_upwards(dist, rank, why):
    """tell the rank that will wait for this rank's hand-over that it is not coming"""
    if rank != 0:
        try:
            _send_bytes(dist, _encode_offer({"kind": "abort", "error": str(why)[:500], "from": rank}), rank - (rank & -rank))
        except Exception:      # (best effort: the process group may already be gone)
            pass


def _refuse_senders(dist, rank, world, first_step):
    """the ranks that would hand over to this rank in rounds first_step, 2·first_step, ...: take their offers and answer "fail", so that they raise
    instead of waiting (an abort marker needs no answer)"""
    st = max(1, first_step)
    while st < world and rank % (2 * st) == 0:
        src2 = rank + st
        if src2 < world:
            try:
                m2, _ = _decode_offer(_recv_bytes(dist, src2))
                if m2["kind"] != "abort":
                    _send_bytes(dist, b"fail", src2)
            except Exception:
                pass
        st *= 2


def tree_abort(rank, world, dist, why):
    """A rank that cannot even ENTER the tree (its own fold raised): the same courtesy as a failed merge — the rank above learns that the hand-over is not
    coming, the ranks below are answered "fail" — so that nobody sits in a blocking receive until the process group's timeout."""
    _abort_upwards(dist, rank, why)
    _refuse_senders(dist, rank, world, 1)


def tree_final_fold(proof, vk, rank, world, dist, merged_cls, shm_dir=None, timings=None):
    """The ranks' final fold as a log-depth pairwise tree ON THE RANKS' OWN GPUs (the north_star's "host-side sequential final fold" with
    the sequence shortened to ceil(log2 N) merges on the critical path): in round k rank r + 2^k hands its merged proof to rank r, which
    folds it in (Node(A, B) of DESIGN.md §6b — one cross term, one large MSM, one fused fold, on r's GPU); pairs start as soon as both
    sides are ready — no barrier.  Hand-over, in order of preference: a HIP IPC ticket (device-to-device copy incl. the running products,
    merged_cls.open_shared), a file in node-local shared memory (save -> load), bytes through gloo.  Returns the proof on rank 0.
    A failure travels UP the tree: a rank whose merge, IPC open or decode fails answers its sender "fail" (which raises there), sends an abort marker to
    the rank that is waiting for ITS hand-over, refuses its later senders, and raises; a rank that receives an abort marker does the same; a sender that
    cannot make its offer (save / share raised) sends the abort marker instead of the offer; a rank whose own FOLD raised never gets here — prove_sharded
    calls tree_abort for it.  Nobody is left in a blocking receive.  (Failures before the digest exchange — a collective — surface at the peers as the
    process group's error when the failing rank exits.)"""
    import os
    import time
    # VIMZ_SHARD_TRANSPORT = ipc | file | bytes pins the hand-over; "ipc-fail" makes the receiver's IPC open fail, to exercise the fallback;
    # "merge-fail:<rank>" makes that rank's merge raise, to exercise the failure path
    forced = os.environ.get("VIMZ_SHARD_TRANSPORT", "")
    fail_rank = int(forced.split(":", 1)[1]) if forced.startswith("merge-fail:") else -1
    can_ipc = hasattr(merged_cls, "open_shared") and (forced in ("", "ipc", "ipc-fail") or fail_rank >= 0)
    t_wait = t_merge = 0.0
    step = 1

    def abort_upwards(why):
        _abort_upwards(dist, rank, why)

    def refuse_later_senders(after_step):
        _refuse_senders(dist, rank, world, 2 * after_step)

    try:
        while step < world:
            if rank % (2 * step) == 0:
                src = rank + step
                if src < world:
                    t0 = time.time()
                    raw = _recv_bytes(dist, src)
                    t1 = time.time()
                    t_wait += t1 - t0
                    try:
                        msg, payload = _decode_offer(raw)
                    except Exception as e:      # (a message that does not parse: its sender IS waiting for an answer)
                        _send_bytes(dist, b"fail", src)
                        abort_upwards(e)
                        refuse_later_senders(step)
                        raise
                    if msg["kind"] == "abort":      # (its sender is not waiting for an answer)
                        err = RuntimeError(f"sharded proof: rank {msg.get('from', src)} failed: {msg.get('error', '')}")
                        abort_upwards(err)
                        refuse_later_senders(step)
                        raise err
                    err = None
                    answer = True      # (False: the sender has told us it gave up and is not waiting for a reply)
                    try:
                        other = None
                        if msg["kind"] == "ipc":
                            try:
                                if forced == "ipc-fail":
                                    raise RuntimeError("IPC open refused (VIMZ_SHARD_TRANSPORT=ipc-fail)")
                                other = merged_cls.open_shared(vk, payload)
                            except Exception:      # no IPC / peer access between the two devices: ask for the bytes instead
                                _send_bytes(dist, b"retry", src)
                                msg, payload = _decode_offer(_recv_bytes(dist, src))
                                if msg["kind"] == "abort":
                                    answer = False
                                    raise RuntimeError(f"sharded proof: rank {msg.get('from', src)} failed: {msg.get('error', '')}")
                        if other is None and msg["kind"] == "file":
                            other = merged_cls.load(vk, np.fromfile(msg["path"], dtype=np.uint8, count=int(msg["bytes"])))
                        elif other is None and msg["kind"] == "bytes":
                            other = merged_cls.load(vk, np.frombuffer(payload, dtype=np.uint8))
                        t_open = time.time()
                        if other is not None:
                            if proof is None:
                                proof = other
                            else:
                                try:
                                    if rank == fail_rank:
                                        raise RuntimeError(f"merge refused (VIMZ_SHARD_TRANSPORT={forced})")
                                    proof.merge(other)
                                finally:
                                    other.close()
                        if timings is not None:
                            timings.setdefault("transports", []).append(msg["kind"])
                            timings.setdefault("hand_overs", []).append({"from": src, "kind": msg["kind"], "open_s": t_open - t1, "merge_s": time.time() - t_open})
                    except Exception as e:      # (the sender must not be left waiting: answer first, then tell the rank above, raise afterwards)
                        err = e
                    if answer:
                        _send_bytes(dist, b"done" if err is None else b"fail", src)
                    if err is not None:
                        abort_upwards(err)
                        refuse_later_senders(step)
                        raise err
                    t_merge += time.time() - t1
            else:
                dst = rank - step
                transport = "ipc" if (can_ipc and proof is not None and hasattr(proof, "share")) else ("bytes" if (forced == "bytes" or not shm_dir) else "file")
                path = None
                if os.environ.get("VIMZ_SHARD_TRANSPORT", "") == f"offer-fail:{rank}":      # (test hook: this rank cannot make its offer)
                    transport = "offer-fail"
                try:
                    try:
                        if transport == "offer-fail":
                            raise RuntimeError("offer refused (VIMZ_SHARD_TRANSPORT=offer-fail)")
                        offer, payload, path = _offer(proof, merged_cls, transport, shm_dir)
                    except Exception as e:      # (the receiver is waiting for an offer: it gets the abort marker instead)
                        _send_bytes(dist, _encode_offer({"kind": "abort", "error": str(e)[:500], "from": rank}), dst)
                        raise
                    _send_bytes(dist, _encode_offer(offer, payload), dst)
                    reply = bytes(_recv_bytes(dist, dst))
                    if reply == b"retry":
                        try:
                            offer, payload, path = _offer(proof, merged_cls, "file" if shm_dir else "bytes", shm_dir)
                        except Exception as e:
                            _send_bytes(dist, _encode_offer({"kind": "abort", "error": str(e)[:500], "from": rank}), dst)
                            raise
                        _send_bytes(dist, _encode_offer(offer, payload), dst)
                        reply = bytes(_recv_bytes(dist, dst))
                    if reply != b"done":
                        raise RuntimeError(f"sharded proof: rank {dst} could not take over this rank's proof")
                finally:
                    if path is not None and os.path.exists(path):
                        os.unlink(path)
                    if proof is not None:
                        proof.close()
                return None
            step *= 2
        return proof
    finally:
        if timings is not None:
            timings["final_fold_s"] = timings.get("final_fold_s", 0.0) + t_merge
            timings["final_fold_wait_s"] = timings.get("final_fold_wait_s", 0.0) + t_wait


def prove_sharded(ivcs, step_inputs, z0, rank=0, world=1, dist=None, timings=None, merged_cls=None, shm_prefix=None, shm_dir=None):
    """ONE proof object of all rows from z0 over `world` ranks (one process per GPU; BASELINE.json north_star: "independent row-folds
    shard embarrassingly across the 8 GPUs ... host-side sequential final fold; no RCCL collectives needed").  Rank r proves the r-th
    contiguous run of rows with fold_segments_merged (len(ivcs) concurrent segments on its GPU); the runs' start states: every rank
    hashes its own rows, the digests are all-gathered, each rank chains over the rows before it (circuits whose digests depend on
    the state: one chain on rank 0, scattered); the ranks' merged proofs are then folded pairwise up a tree on the ranks' own GPUs
    (tree_final_fold: HIP IPC hand-over where the merged class offers it, else node-local shared memory under shm_dir, else gloo).
    Returns the proof on rank 0, None elsewhere.
    timings: state_chain_s (digests + all-gather + chain, plus the local segments' chains; also split as digests_s / allgather_s /
    chain_s), merge_s, final_fold_s (this rank's opening + merging of other ranks' proofs), final_fold_wait_s (waiting for them),
    t_ready / t_done (wall clock: local proof made / this rank's part of the tree finished)."""
    import os
    import time
    if merged_cls is None:
        from .hip import MergedProof as merged_cls
    if shm_dir is None and shm_prefix:
        shm_dir = os.path.dirname(shm_prefix) or "."
    n = len(step_inputs)
    bounds = segment_bounds(n, world)
    z_start = [int(x) for x in z0]
    my_digests = None
    pending = None
    if world > 1:
        t0 = time.time()
        stride = ivcs[0].digest_stride() if hasattr(ivcs[0], "digest_stride") else 0
        if stride:
            # the chain in its two parts: every rank hashes ITS rows (the expensive, state-independent part: one GPU pass, side by
            # side on all GPUs), the digests are exchanged (a few hundred KB), and rank r runs the serial part — two or three small
            # permutations per row on the host — over the rows of the ranks before it.  (Rank 0 hashing everybody's rows on its GPU
            # cost 138 µs per row at 8K: 260 ms before the last of eight ranks could start a 650 ms fold.)
            import torch
            lo, hi = bounds[rank]
            kmax = max(h - l for l, h in bounds)
            mine = np.zeros((kmax, stride, 4), dtype=np.uint64)
            # Where the library hashes rows on the GPU anyway (calls it gives no host-evaluated head batch: long calls, ranks with few host cores), the rank
            # BEGINS its fold now and takes the digests from the folds' own chain passes (MergedProof.fold_segments_begin): its rows are hashed once, and
            # the exchange, the chain over the other ranks' rows and the wait for them all run under the rank's own Poseidon-chain latency.
            # ... only where every rank has a GPU of its own: with two processes on ONE GPU (four hardware queues each) the begun folds' copies and chain passes
            # queue behind the other process's kernels — digests after 18 ms instead of 5, 750 against 947 steps/s for 2 x 256 rows (profiles/r05_pending_ab.txt)
            use_pending = (hi > lo and hasattr(merged_cls, "fold_segments_begin") and all(hasattr(v, "h") for v in ivcs)
                           and os.environ.get("VIMZ_SHARD_NO_PENDING") is None and _head_policy(hi - lo) == 0
                           and (os.environ.get("VIMZ_SHARD_PENDING") == "1" or _own_gpu_per_rank()))
            if use_pending:
                pending = merged_cls.fold_segments_begin(ivcs, step_inputs[lo:hi])
                try:
                    if rank == 0:
                        pending.start(z_start)      # (its state is the proof's z_0: known)
                    mine[:hi - lo] = pending.digests()
                except BaseException:
                    pending.cancel()
                    raise
            elif hi > lo:
                mine[:hi - lo] = np.asarray(ivcs[0].row_digests(step_inputs[lo:hi])).reshape(hi - lo, stride, 4)
                my_digests = mine[:hi - lo]
            t1 = time.time()
            try:
                allg = [torch.empty(mine.size, dtype=torch.int64) for _ in range(world)]
                dist.all_gather(allg, torch.from_numpy(mine.view(np.int64).reshape(-1)))
                t2 = time.time()
                for r in range(rank):
                    plo, phi = bounds[r]
                    if phi > plo:
                        dg = allg[r].numpy().view(np.uint64).reshape(kmax, stride, 4)[:phi - plo]
                        z_start = _ints(ivcs[0].chain_from_digests(z_start, step_inputs[plo:phi], dg)[-1])
                if pending is not None and rank != 0:
                    pending.start(z_start)
            except BaseException:
                if pending is not None:
                    pending.cancel()      # (the begun folds must not outlive a failed exchange)
                raise
            if timings is not None:
                timings["digests_s"] = timings.get("digests_s", 0.0) + t1 - t0
                timings["allgather_s"] = timings.get("allgather_s", 0.0) + t2 - t1
                timings["chain_s"] = timings.get("chain_s", 0.0) + time.time() - t2
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
    try:
        if os.environ.get("VIMZ_SHARD_TRANSPORT", "") == f"fold-fail:{rank}":      # (test hook: this rank's own fold raises)
            if pending is not None:
                pending.cancel()
            raise RuntimeError("fold refused (VIMZ_SHARD_TRANSPORT=fold-fail)")
        if pending is not None:
            proof, tp = pending.finish()
            if timings is not None:
                timings["merge_s"] = timings.get("merge_s", 0.0) + tp["merge_s"]
                timings["pending_fold"] = True
        else:
            proof = fold_segments_merged(ivcs, step_inputs[lo:hi], z_start, timings, merged_cls, digests=my_digests) if hi > lo else None
    except BaseException as e:      # (an unsatisfiable row, a HIP error: the ranks of the tree must not wait for this one — ADVICE r5)
        if world > 1:
            tree_abort(rank, world, dist, e)
        raise
    if timings is not None:
        timings["t_ready"] = time.time()
    if world == 1:
        return proof
    proof = tree_final_fold(proof, ivcs[0], rank, world, dist, merged_cls, shm_dir, timings)
    if timings is not None:
        timings["t_done"] = time.time()
        timings.setdefault("final_fold_s", 0.0)
    return proof


def _own_gpu_per_rank():
    try:
        from . import _lib
        return _lib.default_hw_queues() == 8      # (its rule: at most one rank per visible device)
    except Exception:
        return False


def _head_policy(nsteps):
    try:
        from . import hip
        return hip.head_rows_policy(nsteps, segments=True)
    except Exception:      # (stand-in provers in the CPU tests: no library)
        return 1


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
