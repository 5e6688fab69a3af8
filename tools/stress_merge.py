#!/usr/bin/env python3
"""Repeats fold_input as three merged segments (vimz_ivc_fold_segments) on the same provers: every object must verify, times must
stay flat, and the device memory in use must not grow (the merged proof's buffers are handed back and reused).
usage: stress_merge.py [iterations] [rows]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402
from vimz_amd import folding, hip  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    import torch
    ctxs = [hip.Context(0) for _ in range(3)]
    circuit, params = folding.prepare_folding(ctxs[0], "contrast", "HD")
    rows, z0 = bench.build_inputs("contrast", "HD")
    ivcs = [hip.IVC(c, circuit, params.ck, params.secondary_key(), max_batch=64) for c in ctxs]
    times, mem = [], []
    for it in range(iters):
        lo = (it * 7) % (len(rows) - nrows)
        t0 = time.time()
        m, t = hip.MergedProof.fold_segments(ivcs, rows[lo:lo + nrows], z0)
        dt = time.time() - t0
        code = m.verify(nrows, z0)
        assert code == 0 and m.verify(nrows + 1, z0) != 0, (it, code)
        if it % 10 == 0:
            blob, _ = m.compress()
            assert hip.MergedProof.verify_compressed(ivcs[0], blob, nrows, z0) == 0
        m.close()
        free, total = torch.cuda.mem_get_info(0)
        times.append(dt); mem.append((total - free) / 2**30)
    print(f"{iters} merged proofs of {nrows} rows: steps/s first {nrows / times[0]:.0f}, median {nrows / sorted(times)[len(times) // 2]:.0f}, last {nrows / times[-1]:.0f}; "
          f"device memory in use after iteration 2 / last: {mem[min(2, iters - 1)]:.2f} / {mem[-1]:.2f} GiB")
    assert mem[-1] - mem[min(2, iters - 1)] < 0.25, "device memory grows from proof to proof"
    for v in ivcs:
        v.close()
    params.free()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
