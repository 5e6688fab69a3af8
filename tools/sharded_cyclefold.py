#!/usr/bin/env python3
"""Nova + CycleFold sharded over ranks (one process per GPU; on a one-GPU box the ranks share it): every rank proves its run of rows as S
concurrent segments merged into one object, rank 0 folds the ranks' objects into ONE (vimz_cf_merge_merged) — north_star's "host-side
sequential final fold" for the Sonobe backend's scheme.  Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P tools/sharded_cyclefold.py [rows per rank] [segments]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from vimz_amd import hip, folding  # noqa: E402
from vimz_amd.distributed import prove_sharded  # noqa: E402
from bench import build_inputs  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ndev = torch.cuda.device_count()
    dev = int(os.environ.get("LOCAL_RANK", 0)) % max(1, ndev)
    ctxs = [hip.Context(dev) for _ in range(S)]
    circuit, params = folding.prepare_folding(ctxs[0], "contrast", "HD", backend="sonobe")
    steps, z0 = build_inputs("contrast", "HD")
    z0 = [int(x) for x in z0]
    n = len(steps)
    rows = np.stack([steps[i % n] for i in range(world * K)])            # (image rows reused cyclically when the job is longer than the image)
    cfs = [hip.CycleFoldIVC(c, circuit, params.ck, params.secondary_key(), max_batch=32) for c in ctxs]
    shm = f"/dev/shm/vimz_cf_{os.environ.get('MASTER_PORT', '0')}_"
    warm = prove_sharded(cfs, rows[:world * 8], z0, rank, world, dist if world > 1 else None, merged_cls=hip.CycleFoldMerged, shm_prefix=shm if world > 1 else None)
    if warm is not None:
        warm.close()
    if world > 1:
        dist.barrier()
    tm = {}
    t0 = time.time()
    proof = prove_sharded(cfs, rows, z0, rank, world, dist if world > 1 else None, tm, merged_cls=hip.CycleFoldMerged, shm_prefix=shm if world > 1 else None)
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    if rank == 0:
        print(json.dumps({"metric": "cyclefold_folding_steps_per_sec", "workload": "contrast_step_HD", "n_ranks": world, "gpus_visible": ndev, "rows": world * K, "segments_per_rank": S,
                          "value": world * K / dt, "verified": proof.verify(world * K, z0) == 0, "proof_object": proof.info(),
                          "state_chain_s": tm.get("state_chain_s"), "merge_s": tm.get("merge_s"), "final_fold_s": tm.get("final_fold_s")}))
        proof.close()
    if world > 1:
        dist.barrier()
    for v in cfs:
        v.close()
    params.free()
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
