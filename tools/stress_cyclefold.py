#!/usr/bin/env python3
"""Stress: the same three CycleFold provers fold and merge row segments again and again (fresh process).  Every merged proof must verify,
the rate must not drift and device memory must stay flat.  usage: stress_cyclefold.py [repetitions] [rows]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from vimz_amd import hip, folding  # noqa: E402
from vimz_amd.distributed import fold_concurrently, ivc_segments  # noqa: E402
from bench import build_inputs  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 96
ctxs = [hip.Context(0) for _ in range(3)]
circuit, params = folding.prepare_folding(ctxs[0], "contrast", "HD", backend="sonobe")
steps, z0 = build_inputs("contrast", "HD")
rows = np.stack(steps[:n]); z0 = [int(x) for x in z0]
cfs = [hip.CycleFoldIVC(c, circuit, params.ck, params.secondary_key(), max_batch=32) for c in ctxs]
rates, mem = [], []
for rep in range(reps):
    t = time.time()
    segs = ivc_segments(cfs, rows, z0)
    for v, r, z in segs:
        v.reset(z)
    fold_concurrently([(v, r) for v, r, z in segs])
    m = hip.CycleFoldMerged.of(cfs)
    dt = time.time() - t
    assert m.verify(n, z0) == 0, rep
    m.close()
    rates.append(n / dt)
    free, total = torch.cuda.mem_get_info(0)
    mem.append((total - free) >> 20)
print("steps/s per repetition:", [round(r) for r in rates])
print("device MiB in use:", mem[0], "->", mem[-1])
