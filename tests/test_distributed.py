"""The multi-GPU path on CPU: world_size-2 gloo run of the sharding driver (vimz_amd/distributed.py) over the oracle-backed
stand-in prover, checked against a single-process fold of the same rows."""
import os
import socket
import sys

import numpy as np
import pytest

from vimz_amd.distributed import segment_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_segment_bounds_cover_all_rows():
    for n in (0, 1, 7, 720, 2160):
        for w in (1, 2, 3, 8):
            b = segment_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _get_or_fail(q, procs, timeout):
    """The result a rank put on the queue — or a failure as soon as any rank has died (not after the whole timeout)."""
    import queue
    import time
    t0 = time.time()
    while True:
        try:
            return q.get(timeout=2)
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError(f"a rank exited with {dead} before the result arrived (its traceback is in the captured stderr)")
            if time.time() - t0 > timeout:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError("timed out waiting for the ranks")


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from tests import _oracle
    from tests._oracle_prover import OracleProver
    from tests.test_circuits import step_inputs
    from vimz_amd.circuit import Circuit
    from vimz_amd.distributed import fold_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = _oracle.load()
    c = Circuit.for_resolution("hash", "HD")
    key = orc.seq_bases(0, max(c.n_wires, c.n_constraints))
    z0, inputs = step_inputs("hash")
    rows = np.stack(inputs[:5])
    p = OracleProver(orc, c, key)
    res = fold_sharded(p, rows, z0, rank=rank, world=world, dist=dist)
    if rank == 0:
        q.put((res, _oracle.from_limbs(p.instance()["z"]), p.steps, p.verify()))
    dist.barrier()
    dist.destroy_process_group()


def _gpu_worker(rank, world, port, q):
    """The same driver over the GPU prover (two ranks share the one GPU of the test box: own context, own streams each)."""
    sys.path.insert(0, ROOT)
    os.environ["LOCAL_WORLD_SIZE"] = str(world)      # (both ranks share the one GPU: half the hardware queues per process)
    import torch.distributed as dist
    from tests._oracle import from_limbs
    from tests.test_circuits import step_inputs
    from vimz_amd import _lib, hip
    from vimz_amd.circuit import Circuit
    from vimz_amd.distributed import fold_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = hip.Context(0)
    c = Circuit.for_resolution("hash", "HD")
    ck = ctx.bases_generate(_lib.CURVE_BN254_G1, 1 << 13)
    z0, inputs = step_inputs("hash")
    rows = np.stack(inputs[:7])
    p = hip.Prover(ctx, c, ck, max_batch=2)
    res = fold_sharded(p, rows, z0, rank=rank, world=world, dist=dist)
    if rank == 0:
        inst = p.instance()
        q.put((res, from_limbs(inst["z"]), inst["steps"], p.verify()))
    dist.barrier()
    p.close(); ck.free(); ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_fold_on_the_gpu_prover(oracle):
    """world_size 2 over gloo with the GPU prover behind the driver: rank 1's segment starts at the hash-only state chain's
    state, rank 0 gathers its exported accumulator, merges (host-side final fold) and verifies; the merged chain ends in the
    oracle's state after all seven rows."""
    import torch.multiprocessing as mp
    from tests._oracle import T_HASH
    from tests.test_circuits import step_inputs
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res, z_final, steps, vflags = _get_or_fail(q, procs, 900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert res["verified"] and res["steps"] == 7 and steps == 7 and vflags == 0
    z0, inputs = step_inputs("hash")
    z = list(z0)
    for i in range(7):
        ok, z = oracle.step_eval(T_HASH, z, inputs[i])
    assert z_final == z


def test_two_rank_fold_merges_and_verifies(oracle):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res, z_final, steps, vflags = _get_or_fail(q, procs, 600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res["verified"] and res["steps"] == 5 and steps == 5 and vflags == 0
    # the merged chain ends where a single-process run over the same rows ends
    from tests._oracle import T_HASH
    from tests.test_circuits import step_inputs
    z0, inputs = step_inputs("hash")
    z = list(z0)
    for i in range(5):
        ok, z = oracle.step_eval(T_HASH, z, inputs[i])
    assert z_final == z


def test_merge_of_oracle_provers_satisfies_relaxed_r1cs(oracle):
    """Relaxed+relaxed NIFS (the final fold) keeps Az∘Bz = u·Cz + E and both commitments, for uneven segments."""
    from tests._oracle_prover import OracleProver
    from tests.test_circuits import step_inputs
    from vimz_amd.circuit import Circuit
    c = Circuit.for_resolution("hash", "HD")
    key = oracle.seq_bases(0, max(c.n_wires, c.n_constraints))
    z0, inputs = step_inputs("hash")
    a, b = OracleProver(oracle, c, key), OracleProver(oracle, c, key)
    a.reset(z0); a.fold(inputs[:3])
    b.reset(a.z); b.fold(inputs[3:4])
    with pytest.raises(ValueError):          # segments merge only in row order and only when adjacent
        b.merge(a.export())
    with pytest.raises(ValueError):
        a.merge(a.export())
    a.merge(b.export())
    assert a.verify() == 0 and a.steps == 4
    a.E[0, 0] ^= np.uint64(1)
    assert a.verify() & 1


def test_ivc_segments_chain_their_boundary_states(oracle):
    """IVC mode shards an image as a list of IVC proofs of contiguous row segments (Nova IVC chains cannot be merged): the host
    logic that cuts the rows and finds each segment's start state, over a stand-in whose state chain is the oracle's."""
    from tests._oracle import T_HASH
    from tests.test_circuits import step_inputs
    from vimz_amd.distributed import fold_concurrently, ivc_segments
    z0, inputs = step_inputs("hash")
    rows = np.stack(inputs)

    class Stub:
        def __init__(self): self.z, self.n = None, 0
        def state_chain(self, z_start, rws):
            zs, z = [list(z_start)], list(z_start)
            for r in rws:
                ok, z = oracle.step_eval(T_HASH, z, r)
                assert ok
                zs.append(list(z))
            from tests._oracle import to_limbs
            return np.stack([to_limbs(s) for s in zs])
        def reset(self, z): self.z, self.n = list(z), 0
        def fold(self, rws):
            for r in rws:
                ok, self.z = oracle.step_eval(T_HASH, self.z, r)
                assert ok
                self.n += 1

    stubs = [Stub() for _ in range(3)]
    segs = ivc_segments(stubs, rows, z0)
    assert [len(r) for _, r, _ in segs] == [4, 3, 3] and segs[0][2] == list(z0)
    for s, r, z in segs:
        s.reset(z)
    fold_concurrently([(s, r) for s, r, z in segs])
    for i in range(2):
        assert segs[i][0].z == segs[i + 1][2]            # z_end of segment i = z_start of segment i+1
    single = Stub(); single.reset(z0); single.fold(rows)
    assert segs[2][0].z == single.z and sum(s.n for s in stubs) == 10


# ---- ONE proof object over ranks (prove_sharded): the host logic on CPU over stand-ins, the real thing with two ranks on one GPU --------
class _StubIVC:
    """state_chain / reset / fold over the oracle's step relation (what vimz_amd.hip.IVC offers the sharding driver)."""

    def __init__(self, oracle):
        self.o, self.z, self.z0, self.n = oracle, None, None, 0

    def state_chain(self, z_start, rws):
        from tests._oracle import T_HASH, to_limbs
        zs, z = [list(z_start)], list(z_start)
        for r in rws:
            ok, z = self.o.step_eval(T_HASH, z, r)
            assert ok
            zs.append(list(z))
        return np.stack([to_limbs(s) for s in zs])

    def reset(self, z):
        self.z, self.z0, self.n = list(z), list(z), 0

    def fold(self, rws):
        from tests._oracle import T_HASH
        for r in rws:
            ok, self.z = self.o.step_eval(T_HASH, self.z, r)
            assert ok
            self.n += 1


class _StubIVCDigests(_StubIVC):
    """The same with the chain in its two parts (row digests computed by any rank, serial chain over known digests): the stand-in's
    "digest" of a row is the row itself."""

    def digest_stride(self):
        return 128

    def row_digests(self, rws):
        return np.ascontiguousarray(rws, dtype=np.uint64).reshape(len(rws), -1, 4)[:, :128]

    def chain_from_digests(self, z_start, rws, digests):
        assert (np.asarray(digests).reshape(len(rws), -1, 4) == np.asarray(rws).reshape(len(rws), -1, 4)[:, :128]).all()
        return self.state_chain(z_start, rws)


class _StubMerged:
    """What vimz_amd.hip.MergedProof offers: created from the first segment, merge() of the adjacent next one (an IVC or another
    merged proof), save / load as bytes.  It keeps the op sequence the library keeps (post-order: ("L", z_start, z_end, n) pushes a
    segment, ("N",) folds the two on top), so that a test can replay the TREE the ranks built."""

    def __init__(self, first):
        self.zs, self.ze, self.n, self.segments = list(first.z0), list(first.z), first.n, 1
        self.ops = [("L", tuple(first.z0), tuple(first.z), first.n)]

    def merge(self, nxt):
        if isinstance(nxt, _StubMerged):
            zs, ze, n, k, ops = nxt.zs, nxt.ze, nxt.n, nxt.segments, nxt.ops
        else:
            zs, ze, n, k, ops = nxt.z0, nxt.z, nxt.n, 1, [("L", tuple(nxt.z0), tuple(nxt.z), nxt.n)]
        if list(zs) != self.ze:
            raise ValueError("segments not adjacent")
        self.ze, self.n, self.segments = list(ze), self.n + n, self.segments + k
        self.ops = self.ops + list(ops) + [("N",)]

    def save(self):
        import pickle
        return np.frombuffer(pickle.dumps((self.zs, self.ze, self.n, self.segments, self.ops)), dtype=np.uint8)

    @classmethod
    def load(cls, vk, blob):
        import pickle
        m = cls.__new__(cls)
        m.zs, m.ze, m.n, m.segments, m.ops = pickle.loads(np.asarray(blob, dtype=np.uint8).tobytes())
        return m

    def close(self):
        pass


def _replay_stub_ops(ops):
    """The stand-in's counterpart of tests/_merge.py::replay: evaluates the post-order op sequence, checks adjacency at every node and
    returns (z_start, z_end, n, the tree as nested tuples of segment lengths)."""
    st = []
    for op in ops:
        if op[0] == "L":
            st.append((list(op[1]), list(op[2]), op[3], op[3]))
        else:
            B, A = st.pop(), st.pop()
            assert A[1] == B[0], "a node of the merge tree joins runs that are not adjacent"
            st.append((A[0], B[1], A[2] + B[2], (A[3], B[3])))
    assert len(st) == 1
    return st[0]


def _sharded_worker(rank, world, port, q, shm, digests=False, n_rows=9):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from tests import _oracle
    from tests.test_circuits import step_inputs
    from vimz_amd.distributed import prove_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = _oracle.load()
    z0, inputs = step_inputs("hash")
    rows = np.stack(inputs[:n_rows])
    tm = {}
    proof = prove_sharded([(_StubIVCDigests if digests else _StubIVC)(orc) for _ in range(2)], rows, z0, rank, world, dist, tm, merged_cls=_StubMerged,
                          shm_dir=("/tmp" if shm else None))
    if rank == 0:
        q.put((proof.zs, proof.ze, proof.n, proof.segments, sorted(tm), proof.ops))
    else:
        assert proof is None
    dist.barrier()
    dist.destroy_process_group()


def _run_sharded(world, shm, digests, n_rows):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q, shm, digests, n_rows)) for r in range(world)]
    for p in procs:
        p.start()
    out = _get_or_fail(q, procs, 600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return out


@pytest.mark.parametrize("shm,digests", [(False, False), (True, False), (True, True)])
def test_two_ranks_with_two_segments_each_end_in_one_proof_object(oracle, shm, digests):
    """prove_sharded over gloo, world size 2, two local segments per rank (stand-ins over the oracle's step relation): rank 0's chain
    (or, with the chain in its two parts, every rank's own row digests, all-gathered) gives rank 1 its start state, every rank's segments merge locally, rank 0 folds rank 1's merged proof in — one object about all
    nine rows from z0 that ends where a single chain ends; both ways of moving the bytes."""
    from tests._oracle import T_HASH
    from tests.test_circuits import step_inputs
    zs, ze, n, segments, keys, ops = _run_sharded(2, shm, digests, 9)
    z0, inputs = step_inputs("hash")
    z = list(z0)
    for i in range(9):
        ok, z = oracle.step_eval(T_HASH, z, inputs[i])
    assert (zs, ze, n, segments) == (list(z0), z, 9, 4)
    assert {"final_fold_s", "final_fold_wait_s", "merge_s", "state_chain_s", "t_ready", "t_done"} <= set(keys)
    assert _replay_stub_ops(ops)[3] == ((3, 2), (2, 2))


def _failing_worker(rank, world, port, q, fail_rank, mode="merge-fail"):
    """tree_final_fold with a merge that fails on `fail_rank`: every rank must come back (with an error or, for ranks whose hand-over was taken, None)
    long before the process group's timeout."""
    sys.path.insert(0, ROOT)
    import time
    import torch.distributed as dist
    from tests import _oracle
    from tests.test_circuits import step_inputs
    from vimz_amd.distributed import prove_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["VIMZ_SHARD_TRANSPORT"] = f"{mode}:{fail_rank}"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = _oracle.load()
    z0, inputs = step_inputs("hash")
    rows = np.stack(inputs[:10])
    t0 = time.time()
    try:
        proof = prove_sharded([_StubIVCDigests(orc) for _ in range(2)], rows, z0, rank, world, dist, {}, merged_cls=_StubMerged, shm_dir="/tmp")
        q.put((rank, "returned", proof is None, time.time() - t0))
    except Exception as e:
        q.put((rank, "raised", str(e)[:200], time.time() - t0))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,fail_rank", [(4, 2), (4, 0), (8, 4)])
def test_a_failed_merge_travels_up_the_tree_instead_of_hanging_it(world, fail_rank):
    """a failure in tree_final_fold used to stay inside its pair — the ranks above waited in a blocking receive until the process group's
    timeout.  Now the failing rank answers its sender "fail", tells the rank waiting for it, and refuses the ranks that would hand over to it later."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q, fail_rank)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, what, info, dt = q.get(timeout=240)
        got[r] = (what, info, dt)
    for p in procs:
        p.join(timeout=60)
    assert got[fail_rank][0] == "raised" and "merge refused" in got[fail_rank][1], got
    assert got[0][0] == "raised", got                                     # rank 0 never gets the ONE object — and says so, promptly
    assert got[fail_rank + 1][0] == "raised" and "could not take over" in got[fail_rank + 1][1], got      # the rank whose proof was refused
    assert all(dt < 120 for _, _, dt in got.values()), got              # (gloo's default timeout is 30 minutes)
    # ranks whose hand-over was taken before the failure simply returned None
    assert all(v[0] == "raised" or v[1] is True for v in got.values()), got


@pytest.mark.parametrize("world,fail_rank,mode", [(4, 2, "fold-fail"), (4, 1, "fold-fail"), (8, 0, "fold-fail"), (4, 3, "offer-fail"), (8, 4, "offer-fail")])
def test_a_failed_fold_or_offer_reaches_every_waiting_rank(world, fail_rank, mode):
    """only failures inside a receiver's merge used to be reported.  A rank whose own FOLD raises (an unsatisfiable row, a HIP error) never
    enters the tree, and a sender whose offer cannot be made (save / share raising) sent nothing: the peer sat in a blocking gloo receive until the
    process group's timeout.  Now prove_sharded calls tree_abort for a failed fold and the sender sends the abort marker in place of its offer."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q, fail_rank, mode)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, what, info, dt = q.get(timeout=240)
        got[r] = (what, info, dt)
    for p in procs:
        p.join(timeout=60)
    assert got[fail_rank][0] == "raised" and ("fold refused" if mode == "fold-fail" else "offer refused") in got[fail_rank][1], got
    assert got[0][0] == "raised", got                                     # rank 0 never gets the ONE object — and says so, promptly
    assert all(dt < 120 for _, _, dt in got.values()), got              # (nobody waited for gloo's 30-minute timeout)
    assert all(v[0] == "raised" or v[1] is True for v in got.values()), got
    if mode == "fold-fail":      # the ranks that would have handed over to the failed rank were answered "fail"
        st = 1
        while st < world and fail_rank % (2 * st) == 0:
            if fail_rank + st < world:
                assert got[fail_rank + st][0] == "raised" and "could not take over" in got[fail_rank + st][1], got
            st *= 2


def test_tree_rounds_pair_adjacent_runs():
    from vimz_amd.distributed import tree_rounds
    assert tree_rounds(1) == []
    assert tree_rounds(2) == [[(0, 1)]]
    assert tree_rounds(4) == [[(0, 1), (2, 3)], [(0, 2)]]
    assert tree_rounds(8) == [[(0, 1), (2, 3), (4, 5), (6, 7)], [(0, 2), (4, 6)], [(0, 4)]]
    assert tree_rounds(5) == [[(0, 1), (2, 3)], [(0, 2)], [(0, 4)]]
    assert tree_rounds(6) == [[(0, 1), (2, 3), (4, 5)], [(0, 2)], [(0, 4)]]
    for w in range(1, 20):      # every rank but 0 hands over exactly once; depth = ceil(log2 w)
        rounds = tree_rounds(w)
        assert sorted(snd for rnd in rounds for _, snd in rnd) == list(range(1, w))
        assert len(rounds) == (w - 1).bit_length()


@pytest.mark.parametrize("world,shm,digests,n_rows", [(4, True, True, 10), (4, False, False, 10), (3, True, True, 7), (4, True, True, 3), (8, True, True, 10)])
def test_ranks_fold_their_proofs_pairwise_up_a_tree(oracle, world, shm, digests, n_rows):
    """World size 3, 4 and 8 over gloo: the ranks' merged proofs are folded pairwise up a tree (rank 1 -> 0 and 3 -> 2 side by side, then
    2 -> 0) instead of one after another on rank 0; the op sequence the final object carries IS that tree (replayed here with the
    adjacency check at every node), its statement is all rows from z0 and it ends where a single chain ends.  Also with fewer rows than
    ranks (a rank without rows hands over nothing)."""
    from tests._oracle import T_HASH
    from tests.test_circuits import step_inputs
    from vimz_amd.distributed import segment_bounds
    zs, ze, n, segments, keys, ops = _run_sharded(world, shm, digests, n_rows)
    z0, inputs = step_inputs("hash")
    z = list(z0)
    for i in range(n_rows):
        ok, z = oracle.step_eval(T_HASH, z, inputs[i])
    assert (zs, ze, n) == (list(z0), z, n_rows)
    r_zs, r_ze, r_n, tree = _replay_stub_ops(ops)
    assert (r_zs, r_ze, r_n) == (list(z0), z, n_rows)

    def run(k):      # a rank's own proof: its rows as (at most) two local segments
        b = [hi - lo for lo, hi in segment_bounds(k, 2) if hi > lo]
        return b[0] if len(b) == 1 else tuple(b)
    per_rank = [run(hi - lo) for lo, hi in segment_bounds(n_rows, world) if hi > lo]
    if world == 8:      # (ten rows over eight ranks: 2, 2, 1, 1, 1, 1, 1, 1 — three levels, the shape the driver's eight-GPU run takes)
        q = per_rank
        assert tree == (((q[0], q[1]), (q[2], q[3])), ((q[4], q[5]), (q[6], q[7])))
    elif world == 4 and n_rows == 10:
        assert tree == ((per_rank[0], per_rank[1]), (per_rank[2], per_rank[3]))
    elif world == 3:
        assert tree == ((per_rank[0], per_rank[1]), per_rank[2])
    else:
        assert tree == ((1, 1), 1)
    assert segments == sum(len(r) if isinstance(r, tuple) else 1 for r in per_rank)


def test_fold_segments_merged_handles_fewer_rows_than_segments(oracle):
    from tests.test_circuits import step_inputs
    from vimz_amd.distributed import fold_segments_merged
    z0, inputs = step_inputs("hash")
    m = fold_segments_merged([_StubIVC(oracle) for _ in range(3)], np.stack(inputs[:2]), z0, merged_cls=_StubMerged)
    assert (m.n, m.segments, m.zs) == (2, 2, list(z0))


def _gpu_sharded_worker(rank, world, port, q, oracle_replay=False):
    sys.path.insert(0, ROOT)
    os.environ["LOCAL_WORLD_SIZE"] = str(world)      # (the ranks share the one GPU: vimz_amd/_lib.py halves the hardware queues per process)
    import torch.distributed as dist
    from tests.test_circuits import step_inputs
    from vimz_amd import _lib, hip
    from vimz_amd.circuit import Circuit
    from vimz_amd.distributed import prove_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctxs = [hip.Context(0), hip.Context(0)]
    c = Circuit.for_resolution("hash", "HD")
    ck1 = ctxs[0].bases_generate(_lib.CURVE_BN254_G1, 1 << 14)
    ck2 = ctxs[0].bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-secondary")
    z0, inputs = step_inputs("hash")
    rows = np.stack(inputs[:9])
    ivcs = [hip.IVC(cx, c, ck1, ck2, max_batch=2) for cx in ctxs]
    tm = {}
    proof = prove_sharded(ivcs, rows, z0, rank, world, dist, tm, shm_dir="/tmp")
    if rank == 0:
        replay = None
        if oracle_replay:      # the oracle-side verifier replays the TREE the ranks built (tests/_merge.py) and checks the two folded instances
            from tests import _merge, _oracle
            from tests.test_gpu_ivc import _shape_digest
            orc = _oracle.load()
            failed, acc = _merge.verify_merged(orc, proof, ivcs[0], ck1, ck2, 9, z0, _shape_digest(ivcs[0], 0), _shape_digest(ivcs[0], 1), check_commitments=False)
            kinds = "".join("LN"[k] for k, *_ in _merge.parse_records(proof.records(), c.len_z)["ops"])
            replay = (failed, acc["n"], kinds)
        q.put((proof.verify(9, z0), proof.verify(8, z0), proof.state(), proof.info()["segments"], tm.get("transports"), replay))
        proof.close()
    dist.barrier()
    for v in ivcs:
        v.close()
    ck1.free(); ck2.free()
    for cx in ctxs:
        cx.close()
    dist.destroy_process_group()


def _gpu_sharded_cyclefold_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["LOCAL_WORLD_SIZE"] = str(world)
    import torch.distributed as dist
    from tests.test_circuits import step_inputs
    from vimz_amd import _lib, hip
    from vimz_amd.circuit import Circuit
    from vimz_amd.distributed import prove_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctxs = [hip.Context(0), hip.Context(0)]
    c = Circuit.for_resolution("hash", "HD")
    ck1 = ctxs[0].bases_generate(_lib.CURVE_BN254_G1, 1 << 16)
    ck2 = ctxs[0].bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-cyclefold")
    z0, inputs = step_inputs("hash")
    rows = np.stack(inputs[:9])
    cfs = [hip.CycleFoldIVC(cx, c, ck1, ck2, max_batch=2) for cx in ctxs]
    tm = {}
    proof = prove_sharded(cfs, rows, z0, rank, world, dist, tm, merged_cls=hip.CycleFoldMerged, shm_dir="/tmp")
    if rank == 0:
        q.put((proof.verify(9, z0), proof.verify(8, z0), proof.state(), proof.info()["segments"], tm.get("transports")))
        proof.close()
    dist.barrier()
    for v in cfs:
        v.close()
    ck1.free(); ck2.free()
    for cx in ctxs:
        cx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_the_gpu_end_in_one_verified_cyclefold_proof_object(oracle):
    """The same sharding for the Sonobe backend's scheme: 2 ranks x 2 CycleFold segments -> two merged objects (one run each) -> rank 0's
    final fold (vimz_cf_merge_merged) -> ONE object that verifies for (9 steps, z0) and for nothing else."""
    import torch.multiprocessing as mp
    from tests._oracle import T_HASH
    from tests.test_circuits import step_inputs
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_sharded_cyclefold_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok9, ok8, state, segments, transports = _get_or_fail(q, procs, 900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    z0, inputs = step_inputs("hash")
    z = list(z0)
    for i in range(9):
        ok, z = oracle.step_eval(T_HASH, z, inputs[i])
    assert ok9 == 0 and ok8 != 0 and segments == 4
    assert state == ([int(x) for x in z0], z, 9)
    assert transports == ["ipc"]      # (vimz_cf_merged_share / _open_shared: device-to-device, no host round trip)


def _run_gpu_sharded(world, oracle_replay, transport=None):
    import torch.multiprocessing as mp
    if transport:
        os.environ["VIMZ_SHARD_TRANSPORT"] = transport      # (inherited by the spawned ranks)
    try:
        return _run_gpu_sharded_inner(world, oracle_replay)
    finally:
        os.environ.pop("VIMZ_SHARD_TRANSPORT", None)


def _run_gpu_sharded_inner(world, oracle_replay):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_sharded_worker, args=(r, world, port, q, oracle_replay)) for r in range(world)]
    for p in procs:
        p.start()
    out = _get_or_fail(q, procs, 900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return out


@pytest.mark.gpu
def test_two_ranks_on_the_gpu_end_in_one_verified_proof_object(oracle):
    """world_size 2 over gloo with the GPU provers (both ranks share the one GPU of the box): 2 x 2 IVC segments -> two merged proofs
    -> rank 1 hands its proof to rank 0 by HIP IPC (device-to-device, running products included) -> ONE object that verifies for
    (9 steps, z0) and for nothing else."""
    from tests._oracle import T_HASH
    from tests.test_circuits import step_inputs
    ok9, ok8, state, segments, transports, _ = _run_gpu_sharded(2, False)
    z0, inputs = step_inputs("hash")
    z = list(z0)
    for i in range(9):
        ok, z = oracle.step_eval(T_HASH, z, inputs[i])
    assert ok9 == 0 and ok8 != 0 and segments == 4
    assert state == ([int(x) for x in z0], z, 9)
    assert transports == ["ipc"]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_ranks_that_begin_their_folds_before_their_start_state_is_known(oracle, world):
    """The same with the rank-level deferred start forced on (VIMZ_SHARD_PENDING=1, every row's witness on the GPU): every rank begins its segments' folds,
    takes its row digests from the folds' own chain passes (vimz_ivc_fold_segments_begin), exchanges them, chains over the rows before its own and provides the
    start state; ONE verified object for (9 steps, z0), oracle-side replay of its records included."""
    from tests._oracle import T_HASH
    from tests.test_circuits import step_inputs
    os.environ["VIMZ_SHARD_PENDING"] = "1"; os.environ["VIMZ_HEAD_ROWS"] = "0"      # (inherited by the spawned ranks)
    try:
        ok9, ok8, state, segments, transports, replay = _run_gpu_sharded(world, True)
    finally:
        os.environ.pop("VIMZ_SHARD_PENDING", None); os.environ.pop("VIMZ_HEAD_ROWS", None)
    z0, inputs = step_inputs("hash")
    z = list(z0)
    for i in range(9):
        ok, z = oracle.step_eval(T_HASH, z, inputs[i])
    assert ok9 == 0 and ok8 != 0 and state == ([int(x) for x in z0], z, 9)
    failed, n, kinds = replay
    assert failed == [] and n == 9 and kinds.count("L") == segments      # the oracle-side verifier accepts the tree the ranks built


@pytest.mark.gpu
def test_four_ranks_on_the_gpu_fold_their_proofs_up_a_tree(oracle):
    """world_size 4 (four processes on the one GPU): 4 x 2 IVC segments; rank 1 -> 0 and 3 -> 2 side by side, then 2 -> 0, every
    hand-over by HIP IPC.  The product's verifier accepts the ONE object for (9 steps, z0) only; the oracle-side verifier
    (tests/_merge.py) replays the records — whose op sequence must be exactly that tree — and accepts both folded instances."""
    from tests._oracle import T_HASH
    from tests.test_circuits import step_inputs
    ok9, ok8, state, segments, transports, replay = _run_gpu_sharded(4, True)
    z0, inputs = step_inputs("hash")
    z = list(z0)
    for i in range(9):
        ok, z = oracle.step_eval(T_HASH, z, inputs[i])
    assert ok9 == 0 and ok8 != 0 and segments == 8
    assert state == ([int(x) for x in z0], z, 9)
    assert transports == ["ipc", "ipc"]
    failed, n, kinds = replay
    assert failed == [] and n == 9
    assert kinds == "LLN" "LLN" "N" "LLN" "LLN" "N" "N"      # ((r0 r1) (r2 r3)), every rank's run = two local segments


@pytest.mark.gpu
@pytest.mark.parametrize("transport,seen", [("ipc-fail", ["file"]), ("bytes", ["bytes"])])
def test_ranks_fall_back_to_bytes_where_ipc_is_unavailable(oracle, transport, seen):
    """A node where HIP IPC or peer access between two GPUs is not to be had: the receiving rank's open fails, it asks for the bytes, the
    sender writes them (node-local file with an unpredictable name, or through gloo) and the proof is loaded the round-3 way — the ONE
    object verifies for (9 steps, z0) all the same."""
    ok9, ok8, state, segments, transports, _ = _run_gpu_sharded(2, False, transport)
    assert ok9 == 0 and ok8 != 0 and segments == 4 and state[2] == 9
    assert transports == seen
