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
    res, z_final, steps, vflags = q.get(timeout=900)
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
    res, z_final, steps, vflags = q.get(timeout=600)
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
