#!/usr/bin/env python3
"""Generations of provers on the same contexts: three Nova provers fold a merged proof and are closed, then three more (Nova again, or
Nova + CycleFold) are created on the same three contexts and fold concurrently.  Until the end of round 3 the second generation ran
at 20–250 steps/s instead of 800–930 (a priority inversion between the fold's high-priority streams, stalled in barriers
behind the producer's events, and the producer's low-priority streams; cured by waiting for a row on the host, prover_internal.hpp:
wait_row_flag).  Prints steps/s per fold, the cgroup's throttling counters around each, and the first prover's phases.
usage: prover_generations.py [mode]   mode: (none) Nova then CycleFold | nova2 Nova then Nova | nova2cf Nova, CycleFold, Nova |
       skip CycleFold in a fresh process | nofold provers made and closed without folding first | newctx / keep: new contexts for the second generation"""
import sys, time, os, threading
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench
from vimz_amd import folding, hip
from vimz_amd.distributed import fold_segments_merged

def cg():
    d = {}
    for l in open("/sys/fs/cgroup/cpu.stat"):
        k, v = l.split(); d[k] = int(v)
    return d
def nthreads():
    return len(os.listdir("/proc/self/task"))
def busy_threads(dt=0.2):
    def snap():
        o = {}
        for t in os.listdir("/proc/self/task"):
            try:
                f = open(f"/proc/self/task/{t}/stat").read().rsplit(")", 1)[1].split()
                o[t] = int(f[11]) + int(f[12])
            except Exception: pass
        return o
    a = snap(); time.sleep(dt); b = snap()
    return sorted(((b[t] - a.get(t, 0)) for t in b), reverse=True)[:8]

rows, z0 = bench.build_inputs("contrast", "HD")
rows_a = np.stack(rows)[:96]
ctxs = [hip.Context(0) for _ in range(3)]
mode = sys.argv[1] if len(sys.argv) > 1 else ""
skip_nova = mode == "skip"
if not skip_nova:
    circuit, params = folding.prepare_folding(ctxs[0], "contrast", "HD")
    ck2 = params.secondary_key()
    ivcs = [hip.IVC(c, circuit, params.ck, ck2, max_batch=32) for c in ctxs]
    for rep in range(0 if mode == 'nofold' else 2):
        c0 = cg(); t0 = time.time()
        p = fold_segments_merged(ivcs, rows_a, z0, {})
        dt = time.time() - t0; c1 = cg()
        print(f"nova merged fold {rep}: {96/dt:.0f} steps/s; throttled periods +{c1['nr_throttled']-c0['nr_throttled']}, throttled {1e-6*(c1['throttled_usec']-c0['throttled_usec']):.2f} s, cpu {1e-6*(c1['usage_usec']-c0['usage_usec']):.2f} s in {dt:.2f} s wall; threads {nthreads()}", flush=True)
        assert p.verify(96, z0) == 0
        p.close()
        print("   prover 0 phases, ms:", {k: round(1e3 * v[0], 1) for k, v in ivcs[0].profile().items()}, flush=True)
    if mode != 'keep':
        for i in ivcs: i.close()
        params.free()
    if mode in ('newctx', 'keep'):
        ctxs = [hip.Context(0) for _ in range(3)]
    print("nova provers closed; threads", nthreads(), "busy ticks/0.2s of the top threads while idle:", busy_threads(), flush=True)
if mode.startswith('nova2'):
    if mode == 'nova2cf':      # CycleFold provers first, closed, then Nova provers on the same contexts
        circuit, params = folding.prepare_folding(ctxs[0], "contrast", "HD", backend="sonobe")
        ck2 = params.secondary_key()
        cfs = [hip.CycleFoldIVC(c, circuit, params.ck, ck2, max_batch=32) for c in ctxs]
        for rep in range(2):
            t0 = time.time(); p = fold_segments_merged(cfs, rows_a, z0, {}, merged_cls=hip.CycleFoldMerged); dt = time.time() - t0
            print(f"cyclefold-first merged fold {rep}: {96/dt:.0f} steps/s", flush=True); p.close()
        for c in cfs: c.close()
        params.free()
    circuit, params = folding.prepare_folding(ctxs[0], "contrast", "HD")
    ck2 = params.secondary_key()
    ivcs = [hip.IVC(c, circuit, params.ck, ck2, max_batch=32) for c in ctxs]
    for rep in range(5):
        t0 = time.time(); p = fold_segments_merged(ivcs, rows_a, z0, {}); dt = time.time() - t0
        print(f"nova-again merged fold {rep}: {96/dt:.0f} steps/s", flush=True)
        print("   prover 0 phases, ms:", {k: round(1e3 * v[0], 1) for k, v in ivcs[0].profile().items()}, flush=True)
        assert p.verify(96, z0) == 0
        p.close()
    sys.exit(0)
circuit, params = folding.prepare_folding(ctxs[0], "contrast", "HD", backend="sonobe")
ck2 = params.secondary_key()
cfs = [hip.CycleFoldIVC(c, circuit, params.ck, ck2, max_batch=32) for c in ctxs]
print("cyclefold provers made; threads", nthreads(), "busy:", busy_threads(), flush=True)
prev = None
for rep in range(5):
    c0 = cg(); t0 = time.time()
    p = fold_segments_merged(cfs, rows_a, z0, {}, merged_cls=hip.CycleFoldMerged)
    dt = time.time() - t0; c1 = cg()
    print(f"cyclefold merged fold {rep}: {96/dt:.0f} steps/s; throttled periods +{c1['nr_throttled']-c0['nr_throttled']}, throttled {1e-6*(c1['throttled_usec']-c0['throttled_usec']):.2f} s, cpu {1e-6*(c1['usage_usec']-c0['usage_usec']):.2f} s in {dt:.2f} s wall; threads {nthreads()}", flush=True)
    assert p.verify(96, z0) == 0
    p.close()
    pr = cfs[0].profile()
    cur = {k: v[0] for k, v in pr.items()}
    d = {k: round(1e3 * (cur[k] - (prev[k] if prev else 0)), 1) for k in cur}
    prev = cur
    print("   prover 0 phases, ms in this fold:", d, flush=True)
