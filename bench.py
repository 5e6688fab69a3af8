#!/usr/bin/env python3
"""Benchmark of the Nova folding hot path on MI355X (metric of BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one folding step = one image row through witness generation -> (A,B,C)·z -> MSM(W) -> cross term ->
MSM(T) -> challenge -> fold, on both curves of the cycle.  Workload: contrast_step at HD (the configuration the reference's
headline number is quoted on: 720 steps in 371.7 s = 1.94 steps/s, README.md:52), rows of the reference's sample image
(tests/golden/img2.png, contrast factor 1.4).

  --mode ivc (default)   RecursiveSNARK::prove_step in full: Nova IVC with the augmented verifier circuits on the BN254/Grumpkin
                         cycle (vimz_ivc_*; SURVEY.md §8a incl. rows S1/S2).  `value` is ONE proof object — what the reference's
                         fold_input returns (vimz/src/nova_snark_backend/folding.rs:27-43) — of EXACTLY K timed rows per GPU from the
                         transformation's z0, after a warm-up proof of W rows made the same way; HIP-event profiling off.  The rows
                         are proven as `--segments S` (default 3) contiguous row segments folded concurrently on the GPU and merged
                         into one object (vimz_ivc_merge: out-of-circuit NIFS on both curves); the segments' start states
                         (hash-only chains), the merges and, at N > 1, rank 0's final fold of the ranks' merged proofs are inside the
                         timed region.  The object is verified for exactly (world x K steps, z0) and compressed.  `--segments 1`:
                         one IVC chain.  A second proof of K rows with per-kernel HIP events gives the roofline figure.
  --mode accumulator     the NIFS accumulator over the step circuit's own instances (vimz_prover_*): row segments fold
                         independently and are merged by a host-side final fold (BASELINE.json north_star's sharding picture).
N > 1: every rank folds its own rows (independent row-folds, weak scaling, no data-path collective).  Prints ONE JSON line on
rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Hardware queues of the HIP runtime (read once, when the runtime initialises: set before torch or the library touch the GPU).  The
# default of 4 serialises the fifteen streams of three concurrent segments more than the GPU requires: 8 give 880-890 instead of 784-788
# steps/s over 256 rows — but 8 per process with TWO processes on one GPU collapse to 63 (profiles/r03_hw_queues.txt).  vimz_amd/_lib.py
# picks 8 when a rank has its GPU to itself and the runtime's 4 when ranks share one (LOCAL_WORLD_SIZE over the visible devices); an explicit setting in the environment wins.
from vimz_amd import _lib as _vimz_lib  # noqa: E402,F401  (sets GPU_MAX_HW_QUEUES; touches neither torch nor the GPU)

HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MIXED_ADD_PEAK_GOPS = 15.47      # measured ceiling of the XYZZ mixed addition in the lazily reduced 9x29-bit form (profiles/r02_ubench_fp29.txt)


def build_inputs(transformation, resolution):
    from PIL import Image
    from vimz_amd import folding, image_editor as ie
    img = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "img2.png")).convert("RGB"))
    if resolution != "HD":
        k = {"4K": 3, "8K": 6}[resolution]           # synthetic 4K / 8K: nearest-neighbour upscaling
        img = np.repeat(np.repeat(img, k, axis=0), k, axis=1)
    kw = {"factor": 1.4} if transformation in ("contrast", "brightness") else {}
    if transformation == "resize":
        kw["resize_to"] = {"HD": (640, 480), "4K": (1920, 1080), "8K": (3840, 2160)}[resolution]
    if transformation == "crop":
        kw.update(x=200, y=100, crop_size="SD")      # the reference's Makefile target (Makefile:23)
    inp = ie.build_input(transformation, img, **kw)
    steps, z0 = folding.prepare_input(transformation, inp, resolution)
    return steps, z0


def cpu_baseline(circuit, steps, z0, ck_host, budget_s, threads, ivc=None, ck2_host=None):
    """The oracle (CPU restatement, kind "port") timed on the same workload for about `budget_s` seconds, like for like with what `value`
    counts as a step: per row the step circuit's witness (rows of a batch evaluated side by side on the host threads, after
    the hash-only state chain — the way the GPU path batches them), (A,B,C)·z, MSM(W), cross term, MSM(T) and the five folds of the primary
    instance, AND the recursion's share — the two verifier circuits: their (A,B,C)·z (the oracle's pass over the exported secondary shape,
    once per circuit: 7 925 / 7 912 rows), the four commitments over their wires / rows (two on BN254 G1, two on Grumpkin), their cross terms
    and folds, and the native arithmetic their witness generation is made of (four instance hashes, four 128-bit scalar multiplications:
    oracle/nova.hpp).  Reported beside the GPU number; never the thing measured as `value`.
    Returns (steps/s, seconds, steps, per-phase seconds per step)."""
    from concurrent.futures import ThreadPoolExecutor
    from tests import _oracle
    orc = _oracle.load()
    aux0 = 1 + 2 * circuit.len_z
    key_m = orc.to_mont(1, ck_host.reshape(-1, 4)).reshape(-1, 8)      # Montgomery bases, as the product keeps them
    ph = {"state_chain": 0.0, "witness": 0.0, "spmv": 0.0, "msm_w": 0.0, "cross_term": 0.0, "msm_t": 0.0, "folds": 0.0, "verifier_circuits": 0.0}
    rec = None
    if ivc is not None and ck2_host is not None:      # the recursion's share: shapes and witness vectors of the two verifier circuits as the prover holds them
        from vimz_amd import hip
        tabs2 = ivc.r1cs(1)
        Z2 = np.ascontiguousarray(ivc.export(1, hip.IX_FRESH_Z))
        n2w, n2c = len(Z2), len(tabs2["A_rowptr"]) - 1
        key2_m = orc.to_mont(0, ck2_host.reshape(-1, 4)).reshape(-1, 8)
        rs = np.random.default_rng(11)
        dense = lambda n: np.ascontiguousarray(np.concatenate([rs.integers(0, 1 << 63, size=(n, 3), dtype=np.uint64), rs.integers(0, 1 << 59, size=(n, 1), dtype=np.uint64)], axis=1))
        info = ivc.info()
        nvw, nvc = info["verifier_wires"], info["primary_constraints"] - info["step_constraints"]
        rec = {"tabs2": tabs2, "Z2": Z2, "n2w": n2w, "n2c": n2c, "key2": key2_m, "w1": dense(nvw), "t1": dense(nvc), "t2": dense(n2c), "e2": dense(n2c), "off": info["step_wires"] - 1,
               "g1": (1, 2), "g2": (1, 17631683881184975370165255887551781615748388533673675138860)}
    if len(steps):
        _oracle.witness_execute(orc, circuit, [int(x) for x in z0], steps[0])      # (builds the executor's tables once, before the threads share them)
    t0 = time.time()
    run, n = None, 0
    z = [int(x) for x in z0]
    with ThreadPoolExecutor(max(1, threads)) as ex:
        while n < len(steps) and (n < 2 or time.time() - t0 < budget_s):
            batch = steps[n:n + max(1, threads)]
            t = time.time()
            zs = [z]
            for row in batch:      # hash-only state chain of the batch (serial), then the witnesses side by side
                ok, zn = orc.step_eval(circuit.t, zs[-1], row, *circuit.shape)
                assert ok
                zs.append(zn)
            ph["state_chain"] += time.time() - t; t = time.time()
            wits = list(ex.map(lambda a: _oracle.witness_execute(orc, circuit, a[0], a[1]), zip(zs[:-1], batch)))
            ph["witness"] += time.time() - t
            z = zs[-1]
            for st, w, z_out in wits:
                if time.time() - t0 >= budget_s and n >= 2:
                    break
                t = time.time()
                bad, (a2, b2, c2) = _oracle.r1cs_check(orc, circuit, w, want_products=True, threads=threads)
                assert st == 0 and bad == -1
                ph["spmv"] += time.time() - t; t = time.time()
                _, cW = orc.msm_mont_timed(0, key_m, np.ascontiguousarray(w[aux0:]), threads)
                ph["msm_w"] += time.time() - t
                n += 1
                if run is None:
                    run = [w, a2, b2, c2, np.zeros_like(a2), 1]
                else:
                    t = time.time()
                    T = orc.cross_term(0, run[1], run[2], run[3], run[5], a2, b2, c2, 1, threads=threads)
                    ph["cross_term"] += time.time() - t; t = time.time()
                    _, cT = orc.msm_mont_timed(0, key_m, T, threads)
                    ph["msm_t"] += time.time() - t; t = time.time()
                    r = (cT[0] ^ cW[0]) & ((1 << 128) - 1)        # any 128-bit challenge: the arithmetic cost does not depend on it
                    run = [orc.axpy(0, run[0], r, w, threads), orc.axpy(0, run[1], r, a2, threads), orc.axpy(0, run[2], r, b2, threads), orc.axpy(0, run[3], r, c2, threads),
                           orc.axpy(0, run[4], r, T, threads), (run[5] + r) % orc.modulus[0]]
                    ph["folds"] += time.time() - t
                if rec is not None:
                    t = time.time()
                    r128 = (1 << 128) | (cW[0] & ((1 << 128) - 1))
                    for _ in range(2):      # (A,B,C)·z of a verifier circuit: once per circuit
                        orc.r1cs_check_relaxed(1, rec["tabs2"], rec["n2w"], rec["Z2"], u=1, threads=threads)
                    orc.msm_mont_timed(0, key_m[rec["off"]:], rec["w1"], threads); orc.msm_mont_timed(0, key_m[circuit.n_constraints:], rec["t1"], threads)
                    orc.msm_mont_timed(1, rec["key2"], np.ascontiguousarray(rec["Z2"][1:rec["n2w"] - 2]), threads); orc.msm_mont_timed(1, rec["key2"], rec["t2"], threads)
                    for fid in (0, 1):
                        T2 = orc.cross_term(fid, rec["t2"], rec["e2"], rec["t2"], 3, rec["e2"], rec["t2"], rec["e2"], 1)
                        for _ in range(5):
                            orc.axpy(fid, rec["e2"], r128 % orc.modulus[fid], T2)
                    for cid, g in ((0, rec["g1"]), (1, rec["g2"])):      # NIFS.V inside the circuits: W + rho·W', E + rho·T on the other curve
                        orc.curve_mul(cid, g, r128); orc.curve_mul(cid, g, r128 ^ 5)
                    for fid in (0, 1):
                        orc.nova_instance_hash(fid, 7, n, [1], [2], [1, 2, 1, 2, 1, 3, 4]); orc.nova_instance_hash(fid, 7, n + 1, [1], [3], [1, 2, 1, 2, 2, 3, 4])
                    ph["verifier_circuits"] += time.time() - t
    dt = time.time() - t0
    return n / dt, dt, n, {k: v / max(1, n) for k, v in ph.items()}


def memory_now(torch, device, dist=None):
    """The peak-memory column of the reference's table (README.md:48-56; benchmark.sh:4-8 reads it from /usr/bin/time): device memory in
    use on this rank's GPU right after the timed job (the provers keep every buffer they ever needed until they are closed, so in use
    now = peak; all processes on the device count) and this process's peak resident set."""
    import resource
    dev = None
    try:
        free, total = torch.cuda.mem_get_info(device)
        dev = int(total - free)
    except Exception:
        pass
    rss = int(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss) * 1024
    return {"device_bytes": dev, "host_rss_bytes": rss, "note": "rank 0's GPU and process; device: in use right after the timed job (buffers are kept until close, other processes on the GPU included); host: ru_maxrss"}


def usable_cores():
    """Host cores this process may actually use: the CPU affinity mask capped by the cgroup's CPU quota (the GPU boxes show 256 logical
    CPUs and grant 16: threads beyond the quota only get the whole group throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    # the ranks of a node share the quota: a rank's share (a launcher that binds every rank to cores of its own has already narrowed the mask;
    # --cores does so here)
    local = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    bound = hasattr(os, "sched_getaffinity") and len(os.sched_getaffinity(0)) < (os.cpu_count() or 1)
    if local > 1 and not bound:
        n = max(1, n // local)
    return n


def _ints(limbs):
    return [int(a[0]) | int(a[1]) << 64 | int(a[2]) << 128 | int(a[3]) << 192 for a in limbs]


def _kernel_source_sha():
    """sha256[:16] of the MSM kernels' source: profiles/pmc_traffic.json records it with every entry, so a counter figure measured on
    other kernel code is flagged stale instead of being quoted as current."""
    import hashlib
    h = hashlib.sha256()
    for f in ("msm.hpp", "ec.hpp", "fp29.hpp"):
        with open(os.path.join(ROOT, "vimz_amd", "csrc", f), "rb") as fp:
            h.update(fp.read())
    return h.hexdigest()[:16]


def _pmc_traffic(key):
    """HBM bytes per launch of the roofline kernel from SEPARATE rocprofv3 --pmc passes of this command (the counters cannot be
    read from inside the process): profiles/pmc_traffic.json, written by tools/update_pmc_traffic.py from tools/refresh_profiles.sh's
    passes; the entry names its run.  Returns (bytes, source, stale): stale = the kernels' source has changed since the passes."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fp:
            e = json.load(fp).get(key, {})
        return e.get("hbm_bytes_per_launch"), e.get("source"), (e.get("kernel_source_sha") != _kernel_source_sha()) if e else None
    except OSError:
        return None, None, None


def main_ivc(args, rank, world, dist, torch, ctxs, circuit, params, steps_all, glob, lo, hi, mine, z0, t_setup):
    """Nova IVC (the default mode).  `value`: ONE proof object of world x K rows from the transformation's z0.  Every rank proves its K
    rows as S contiguous segments folded concurrently (S = 3 by default: one chain leaves a quarter of an MI355X idle) and merged
    (vimz_ivc_merge); at N > 1 rank 0 then performs the host-side sequential final fold of the ranks' merged proofs.  The segment
    start states (hash-only chains), the merges and the final fold are all inside the timed region."""
    from vimz_amd import _lib, hip
    from vimz_amd.distributed import fold_segments_merged, prove_sharded
    S = len(ctxs)
    W, K = args.warmup, args.steps
    ck2 = params.secondary_key()
    ivcs = args.made_provers or [hip.IVC(c, circuit, params.ck, ck2, max_batch=args.batch) for c in ctxs]
    helper_ctxs, helper_keys = [], []
    for hidx in range(args.msm_helpers):      # one proof on several GPUs — the large MSM(T) split by base range
        ndev = max(1, torch.cuda.device_count())
        dev = (ctxs[0].device + 1 + hidx) % ndev
        hc = hip.Context(dev)
        hk = params.ck if dev == ctxs[0].device else hc.bases_generate(_lib.CURVE_BN254_G1, params.ck.n)
        ivcs[0].add_msm_helper(hc, hk)
        helper_ctxs.append(hc); helper_keys.append(hk)
    setup_s = time.time() - t_setup
    z0 = [int(x) for x in z0]
    rows_warm, rows_timed, rows_prof = mine[:W], mine[W:W + K], mine[W + K:W + 2 * K]
    if os.environ.get("VIMZ_BENCH_REGISTER_ROWS"):      # experiment (the slow pass): the rows in page-locked memory, so that their upload is a plain DMA
        import ctypes
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
        print(f"[bench] hipHostRegister(rows) -> {_hip.hipHostRegister(mine.ctypes.data, mine.nbytes, 0)}", file=sys.stderr)
    # `value` is the ALL-HIP schedule: every row's witness — Poseidon chains included — on the GPU, whatever the length of the call.  The
    # library's default policy evaluates the chains of a SHORT call's rows (proofs of <= 28 rows: the driver's 20-row window, no real image) on a host pool
    # while the GPU does the rest; that hybrid is reported as an extra (`host_head_batch_schedule`), or becomes `value` with --host-head-batch.
    if not args.host_head_batch:
        hip.set_head_rows(0)

    def sync_all():
        for c in ctxs:
            c.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def prove(rows, z_start, timings=None):
        """one proof object of `rows` from z_start on this rank"""
        return fold_segments_merged(ivcs, rows, z_start, timings)

    # warm-up: a proof of W rows, made and dropped exactly like the timed one (first-call allocations, pinned buffers, thread pools)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"      # (only used where the ranks cannot hand their proofs over by HIP IPC)
    sharded = world > 1 and not args.proof_set
    if W:      # (N > 1: through the same sharded path — digests exchange, tree of hand-overs — so that connections and mappings exist)
        if sharded:
            warm_all = np.ascontiguousarray(steps_all[[glob[r * (W + 2 * K) + i] for r in range(world) for i in range(W)]])
            pw = prove_sharded(ivcs, warm_all, z0, rank, world, dist, shm_dir=shm)
        else:
            pw = prove(rows_warm, z0)
        if pw is not None:
            pw.close()
    # The timed job, R times (--repeats, default 7): each pass is ONE proof of world x K rows from z0 (proof sets: every rank its own proof of
    # K rows), bracketed by barrier + synchronise on both sides; `value` is computed from the MEDIAN pass, every pass's time is printed.
    # Rank r's rows start at the state after the r·K rows of the ranks before it: rank 0 runs that hash-only chain once and hands every
    # rank its start state; the ranks' merged proofs go to rank 0 through node-local shared memory and are folded in row order.
    R = max(1, args.repeats)
    passes = []
    timed_all = None
    if sharded:
        timed_all = np.ascontiguousarray(steps_all[[glob[r * (W + 2 * K) + W + i] for r in range(world) for i in range(K)]])
    for rep in range(R):
        tm = {}
        sync_all()
        if world == 1 and rep == R - 1:
            ctxs[0].trace_marker(1)      # (an empty kernel in a profiler's trace: tools/trace_busy.py cuts the LAST timed pass out between markers 1 and 2)
        cpu0 = time.process_time()      # (user + system time of every thread of this process: what the timed job costs the host)
        t0 = time.time()
        if sharded:
            proof = prove_sharded(ivcs, timed_all, z0, rank, world, dist, tm, shm_dir=shm)
        else:
            proof = prove(rows_timed, z0, tm)
            tm["t_ready"] = tm["t_done"] = time.time()
        sync_all()
        dt = time.time() - t0
        host_cpu_s = time.process_time() - cpu0
        if world == 1 and rep == R - 1:
            ctxs[0].trace_marker(2)
        if dist is not None:      # the slowest rank's time is the pass's time
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        # acceptance of every pass: its ONE object verifies for exactly (world x K steps, z0)   [proof sets: every rank's own proof for (K, z0)]
        code = None
        if proof is not None:
            n_claim = K if (args.proof_set or world == 1) else world * K
            code = proof.verify(n_claim, z0)
        passes.append({"dt": dt, "host_cpu_s": host_cpu_s, "tm": tm, "t0": t0, "verify_code": code, "prof": [ivc.profile() for ivc in ivcs]})
        if rep < R - 1 and proof is not None:
            proof.close()
    order = sorted(range(R), key=lambda i: passes[i]["dt"])
    med = passes[order[R // 2]]
    dt, host_cpu_s, tm, t0 = med["dt"], med["host_cpu_s"], med["tm"], med["t0"]
    samples_ms = [1e3 * p_["dt"] / max(1, K) for p_ in passes]
    all_passes_verified = all(p_["verify_code"] in (0, None) for p_ in passes)
    mem = memory_now(torch, ctxs[0].device)
    state_chain_s, final_fold_s = tm.get("state_chain_s", 0.0), tm.get("final_fold_s", 0.0)
    prof1 = med["prof"]
    # the tail the ranks' final fold adds to the job: from the moment the LAST rank has its own proof to the moment rank 0 holds the one
    # object (the ranks of a node share the wall clock); the prologue: from the start to the moment the LAST rank knows its start state
    t_ready_max, t_done_0, prologue_max = tm["t_ready"], tm["t_done"], tm.get("state_chain_s", 0.0)
    tree = None
    if dist is not None:
        t = torch.tensor([dt, tm["t_ready"], prologue_max, tm.get("digests_s", 0.0), tm.get("allgather_s", 0.0), tm.get("chain_s", 0.0),
                          tm.get("final_fold_s", 0.0), tm.get("final_fold_wait_s", 0.0)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, t_ready_max, prologue_max = float(t[0]), float(t[1]), float(t[2])
        tree = {"hand_overs_to_rank0": tm.get("hand_overs"), "digests_s_max": float(t[3]), "allgather_s_max": float(t[4]), "chain_s_max": float(t[5]), "merging_s_max_over_ranks": float(t[6]),
                "waiting_for_partner_s_max_over_ranks": float(t[7])}
    if os.environ.get("VIMZ_BENCH_DEBUG"):      # every rank's own view of the timed job
        print(f"[bench rank {rank}] " + json.dumps({k: (round(v - t0, 4) if k in ("t_ready", "t_done") else v) for k, v in tm.items()}), file=sys.stderr, flush=True)
    final_tail_s = max(0.0, t_done_0 - t_ready_max) if world > 1 else 0.0
    t_fold = dt - final_tail_s
    timed_rows = K
    timed_total = K * world
    # acceptance: the ONE object verifies for exactly (world x K steps, z0)   [proof sets: every rank's own proof for (K, z0)]
    t_v = time.time()
    ok, codes = True, []
    if proof is not None:
        n_claim = K if (args.proof_set or world == 1) else world * K
        code = proof.verify(n_claim, z0)
        codes = [code]
        ok = code == 0 and proof.verify(n_claim + 1, z0) != 0 and all_passes_verified
    verify_s = time.time() - t_v
    merged_info = proof.info() if proof is not None else None
    merge_prof = proof.profile() if proof is not None else None
    # CompressedSNARK::prove / verify of the merged proof (mod.rs:52-67; README.md:196 counts it into the total proof time)
    compress = None
    if rank == 0 and proof is not None and not args.no_compress:
        try:
            blob, tc = proof.compress()
            t_cv = time.time()
            code_c = hip.MergedProof.verify_compressed(ivcs[0], blob, merged_info["steps"], z0)
            compress = {"setup_s": tc["setup_s"], "prove_s": tc["prove_s"], "verify_s": time.time() - t_cv, "proof_bytes": int(len(blob)), "verified": code_c == 0,
                        "arguments": "one for the folded primary and one for the folded secondary instance of the merged proof"}
            ok = ok and code_c == 0
        except Exception as e:
            print(f"[bench] compression skipped: {e}", file=sys.stderr)
    if proof is not None:
        proof.close()
    # second pass, outside the timed region: the same K rows proven again with HIP events around every kernel of the primary MSM(T)
    # launches on the stream they run on -> the roofline figure
    for c in ctxs:
        c.set_profiling(True)
        c.msm_profile_totals(reset=True)
    t1 = time.time()
    p2 = prove(rows_prof if len(rows_prof) == K else rows_timed, z0)
    for c in ctxs:
        c.sync()
    dt_prof = time.time() - t1
    p2.close()
    tots = [c.msm_profile_totals() for c in ctxs]
    tot = {"ms": {k: sum(t["ms"][k] for t in tots) for k in tots[0]["ms"]}, "calls": sum(t["calls"] for t in tots),
           "points": sum(t["points"] for t in tots), "entries": sum(t["entries"] for t in tots)}
    for c in ctxs:
        c.set_profiling(False)
    if dist is not None:
        t = torch.tensor([1 if ok else 0], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(int(t[0]))
    if rank == 0:
        info = ivcs[0].info()
        n_w, n_c, nnz = info["primary_wires"], info["primary_constraints"], info["primary_nnz"]
        calls = max(1, tot["calls"])
        acc_ms = tot["ms"]["accumulate"] / calls
        msm_ms = sum(tot["ms"].values()) / calls
        alg_bytes = 96.0 * tot["points"] / calls       # 32 B scalar + 64 B base per point of the launch
        achieved = alg_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms else 0.0
        adds = tot["entries"] / calls
        # for reference: the same kernel over the same kind of data (the running error vector of the first segment's proof, which has
        # the cross terms' zero and repetition structure) with nothing else on the GPU
        alone_ms = None
        try:
            E = np.ascontiguousarray(ivcs[0].export(0, hip.IX_RUNNING_E))[:info["step_constraints"]]
            vec = ctxs[0].vec_from_host(_lib.FIELD_BN254_FR, E)
            ctxs[0].set_profiling(True)
            ms = []
            for _ in range(7):
                ctxs[0].msm_vec(params.ck, vec)
                ms.append(ctxs[0].msm_last_profile()["ms"]["accumulate"])
            ctxs[0].set_profiling(False)
            vec.free()
            alone_ms = sorted(ms[1:])[len(ms[1:]) // 2]
        except Exception as e:                          # informational only
            print(f"[bench] isolated k_accum measurement skipped: {e}", file=sys.stderr)
        # the kernel on a step's CRITICAL chain: the fused small MSM (four per step over the verifier circuits' 7.7 k wires / rows),
        # alone on the GPU over dense scalars of that size (in a fold it runs under the bulk kernels: profiles/ has the in-bench trace)
        small = None
        try:
            n_s = info["verifier_wires"] - 2
            rs = np.random.default_rng(5)
            dense = rs.integers(0, 1 << 63, size=(n_s, 4), dtype=np.uint64); dense[:, 3] &= np.uint64((1 << 60) - 1)
            vec = ctxs[0].vec_from_host(_lib.FIELD_BN254_FR, dense)
            ctxs[0].set_profiling(True)
            ms = []
            for _ in range(9):
                ctxs[0].msm_vec(params.ck, vec, base_offset=info["step_wires"] - 1)
                ms.append(ctxs[0].msm_last_profile()["ms"]["accumulate"])
            ctxs[0].set_profiling(False)
            vec.free()
            sm = sorted(ms[1:])[len(ms[1:]) // 2]
            k_win = 37
            small = {"kernel": "k_msm_small (one launch: digits, LDS counting sort, sub-bucket sums, segment tree, chunk merge, weighted bucket reduction)",
                     "points": n_s, "windows": k_win, "algorithmic_bytes_per_launch": 96.0 * n_s, "kernel_ms_alone_on_gpu": sm,
                     "achieved": 96.0 * n_s / (sm * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": 96.0 * n_s / (sm * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "bucket_additions": n_s * k_win, "int_utilisation": n_s * k_win / (sm * 1e-3) / 1e9 / MIXED_ADD_PEAK_GOPS,
                     "bound": "latency: 6 dependent mixed additions, then 19 four-lane levels of dependent additions at ~3.6 us on one wave per SIMD (DESIGN.md §4, §9c); four launches per step, two of them in series"}
        except Exception as e:
            print(f"[bench] isolated k_msm_small measurement skipped: {e}", file=sys.stderr)
        traffic, traffic_src, traffic_stale = _pmc_traffic(f"{args.transformation}_step_{args.resolution}_ivc")
        n_w2, n_c2, nnz2 = info["secondary_wires"], info["secondary_constraints"], info["secondary_nnz"]
        step_bytes = sum(96 * w + 96 * c + 8 * z + 32 * w + 96 * c + 7 * 32 * c + 3 * 32 * w + 12 * 32 * c for w, c, z in ((n_w, n_c, nnz), (n_w2, n_c2, nnz2)))
        phases = {k: 1e3 * sum(p1[k][0] for p1 in prof1) / max(1, timed_rows) for k in prof1[0]}      # (reset() at the start of the timed proof zeroed the counters)
        # extra (N = 1): the same K rows as ONE chain (one IVC, no segments, no merge) — what a single sequential prove_step loop reaches
        one_chain = None
        if world == 1 and S > 1 and not args.no_extras:
            # a chain of its OWN — a context and a prover made for it, the way `--segments 1` runs — not one of the three segments' provers: the streams of
            # provers that were created side by side are interleaved over the hardware queues, which suits the three-segment schedule (+5 %) and costs a lone
            # chain 8 % (profiles/r05_setup_ab.txt)
            c1 = iv1 = None
            try:
                c1 = hip.Context(ctxs[0].device)
                iv1 = hip.IVC(c1, circuit, params.ck, ck2, max_batch=args.batch)
                ds = []
                for rep in range(4):      # the first pass warms the prover's buffers and thread pools
                    iv1.reset(z0)
                    c1.sync()
                    t3 = time.time()
                    iv1.fold(rows_timed)
                    c1.sync()
                    if rep:
                        ds.append(time.time() - t3)
                one_chain = {"steps_per_s": K / sorted(ds)[len(ds) // 2], "samples_steps_per_s": [K / d for d in ds], "verified": iv1.verify(K, z0) == 0,
                             "note": "the same rows as ONE IVC chain on a prover of its own (what bench.py --segments 1 measures); median of three passes"}
            except Exception as e:
                print(f"[bench] one-chain extra skipped: {e}", file=sys.stderr)
            finally:
                if iv1 is not None:
                    iv1.close()
                if c1 is not None:
                    c1.close()
        # extra (N = 1): the same timed passes under the library's DEFAULT policy for short calls — the first rows' Poseidon chains on a host pool (a hybrid:
        # only proofs of <= 28 rows take it) — where `value` was measured with every row's witness on the GPU
        all_hip = None
        if world == 1 and not args.no_extras and not args.host_head_batch:
            try:
                hip.set_head_rows(-1)
                if hip.head_rows_policy(K, segments=S > 1) > 0:
                    pw = prove(rows_timed, z0); pw.close()      # (first use of this schedule's buffers)
                    ds = []
                    for _ in range(R):
                        sync_all()
                        t4 = time.time()
                        pa = prove(rows_timed, z0)
                        sync_all()
                        ds.append(time.time() - t4)
                        ok_a = pa.verify(K, z0) == 0
                        pa.close()
                    all_hip = {"steps_per_s": K / sorted(ds)[R // 2], "samples_steps_per_s": [K / d for d in ds], "verified": bool(ok_a), "head_rows": int(ivcs[0].info().get("head_rows", -1)),
                               "note": "the same passes with the library's default policy (vimz_set_head_rows(-1)): a proof this short has its rows' Poseidon chains evaluated on a host "
                                       "pool while the GPU does the rest — a hybrid no whole image takes; `value` is the all-HIP schedule"}
            except Exception as e:
                print(f"[bench] host-head-batch extra skipped: {e}", file=sys.stderr)
            finally:
                hip.set_head_rows(0)
        # extra (N = 1): the reference's second backend on the same image — Nova + CycleFold (vimz_cf_*): the whole image as one
        # chain and as ONE merged proof of S concurrent segments (vimz_cf_merge), each in a process of its own the way `vimz -b sonobe` would run
        # (tools/e2e.py; a child process, started the ordinary way — this one keeps its GPU state)
        sonobe = None
        if world == 1 and not args.no_extras and not args.proof_set and args.resolution == "HD":
            try:
                import subprocess
                runs = {}
                for segs in sorted({1, S}):
                    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e.py"), args.transformation, args.resolution, str(segs), "cyclefold"],
                                       cwd=ROOT, capture_output=True, text=True, timeout=600)
                    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                    if r.returncode != 0 or not line:
                        raise RuntimeError(f"tools/e2e.py exited with {r.returncode}: {r.stderr[-300:]}")
                    runs[segs] = json.loads(line[-1])
                one = runs[1]
                sonobe = {"one_chain_steps_per_s": one["steps_per_s"], "steps": one["steps"], "verified": bool(one["verified"]), "fold_s": one["spans_s"]["Fold input"],
                          "main_constraints": one["info"]["main_constraints"], "cyclefold_constraints": one["info"]["cyclefold_constraints"],
                          "one_chain_ms_per_step": one["ms_per_step_first_segment"], "decider": one.get("decider"),
                          "note": "Nova + CycleFold IVC (the Sonobe backend's prove_step loop, vimz/src/sonobe_backend/folding.rs:52-66) over the whole image, same kernels; own process"}
                if S > 1:
                    mg = runs[S]
                    sonobe.update({"merged_steps_per_s": mg["steps_per_s"], "segments": S, "merged_verified": bool(mg["verified"]), "merged_fold_s": mg["spans_s"]["Fold input"],
                                   "state_chain_s": mg["state_chain_s"], "merge_s": mg["merge_s"]})
            except Exception as e:
                print(f"[bench] sonobe-backend extra skipped: {e}", file=sys.stderr)
        try:
            fingerprint = ctxs[0].host_fingerprint()
        except Exception as e:                          # informational only
            fingerprint = {"error": str(e)}
        head_rows = int(info.get("head_rows", -1)) if isinstance(info, dict) else -1
        out = {
            "metric": "nova_folding_steps_per_sec",
            "value": timed_total / dt,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": timed_rows,
            "warmup": W,
            "ms_per_step": dt / max(1, timed_rows) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (256-bit Montgomery integers over BN254 Fr/Fq)",
            "data": "synthetic: rows of the reference sample image img2.png, contrast factor 1.4" + ("" if args.resolution == "HD" else f", upscaled to {args.resolution}"),
            "config": {"workload": f"{args.transformation}_step_{args.resolution}", "mode": "ivc (augmented circuits on BN254/Grumpkin: RecursiveSNARK::prove_step in full)",
                       "constraints": n_c, "wires": n_w, "nnz": nnz, "step_circuit_constraints": info["step_constraints"], "step_circuit_wires": info["step_wires"],
                       "secondary_constraints": n_c2, "secondary_wires": n_w2,
                       "rows_per_rank": timed_rows, "segments_per_gpu": S, "witness_batch": args.batch, "msm_helper_contexts": args.msm_helpers,
                       "parallelism": (f"proof set {args.proof_set}: rank r proves proof_set[r % len]; independent proofs, replicas only" if args.proof_set else
                                       (f"ONE proof object: {S} contiguous row segments per GPU folded concurrently as Nova IVCs and merged (vimz_ivc_merge)" if S > 1 else "one IVC chain per GPU") +
                                       ("" if world == 1 else f"; {world} GPUs prove {world} contiguous runs of rows, rank 0 folds their merged proofs into ONE object (host-side sequential final fold); no data-path collective"))},
            "timed_passes": R,
            "value_is": f"median of {R} timed passes of exactly {timed_rows} rows per GPU, each ONE proof from z0 bracketed by barrier + synchronise; every pass's proof verified",
            "samples_ms_per_step": samples_ms, "samples_steps_per_s": [timed_total / (m * 1e-3 * max(1, timed_rows)) for m in samples_ms],
            "samples_min_max_steps_per_s": [timed_total / (max(samples_ms) * 1e-3 * max(1, timed_rows)), timed_total / (min(samples_ms) * 1e-3 * max(1, timed_rows))],
            "samples_spread_pct": 100.0 * (max(samples_ms) - min(samples_ms)) / (sorted(samples_ms)[len(samples_ms) // 2]),
            "head_rows": head_rows,
            "host_fingerprint": fingerprint,
            "verified": bool(ok),
            "verify_codes": codes,
            "proof_object": merged_info,
            "folded_steps_total": merged_info["steps"] if merged_info else None,
            "verify_s": verify_s,
            "state_chain_s": state_chain_s,
            "merge_s": tm.get("merge_s", 0.0),
            "final_fold_s": final_tail_s if world > 1 else 0.0,
            "final_fold_rank0_merging_s": final_fold_s,
            "prologue_s_max_over_ranks": prologue_max if world > 1 else 0.0,
            "sharding": tree,
            "gpus_shared": bool(getattr(args, "gpus_shared", False)),
            "host_cores_per_rank": usable_cores(),
            "fold_s": t_fold,
            "peak_device_bytes": mem["device_bytes"], "peak_host_rss_bytes": mem["host_rss_bytes"], "memory_note": mem["note"],
            "host_cpu": {"core_seconds_rank0": host_cpu_s, "cores_busy_rank0": host_cpu_s / dt, "core_ms_per_step_rank0": 1e3 * host_cpu_s / max(1, K),
                         "note": "process CPU time (all threads: folding threads, helpers, issuers, pools, HIP runtime) of rank 0 inside the timed region"},
            "merge_profile_s": merge_prof,
            "one_chain": one_chain,
            "host_head_batch_schedule": all_hip,
            "sonobe_backend": sonobe,
            "compressed_snark": compress,
            "end_to_end_estimate_s": {"keygen_and_setup": setup_s, "setup_split": getattr(args, "setup_split", None), "fold_720_steps_one_gpu": 720 * dt / max(1, timed_total),
                                      "compress": (compress["setup_s"] + compress["prove_s"]) if compress else None,
                                      "total_720_steps": (setup_s + 720 * dt / max(1, timed_total) + compress["setup_s"] + compress["prove_s"]) if compress else None,
                                      "reference_cpu_server": {"keygen_s": 6.5, "fold_s": 371.7, "compress_s_sample_run": 13.0, "source": "README.md:52, sample-output.png"}},
            "published_reference": {"contrast_HD_steps_per_s_cpu_server": 1.94, "source": "README.md:52 (720 steps / 371.7 s)"},
            "phase_ms_per_step_per_proof": phases,
            "roofline_critical_chain_kernel": small,
            "roofline": {"bound": "hbm", "kernel": "k_accum (bucket accumulation) of the primary MSM(T) launches over the step circuit's rows",
                         "measured": f"HIP events on the kernel's own stream over a second proof of {timed_rows} rows made the same way right after the timed region (events off while `value` is timed); that pass ran at {timed_rows / dt_prof:.1f} steps/s",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": acc_ms, "launches": tot["calls"], "msm_gpu_ms": msm_ms,
                         "mixed_adds_per_launch": adds, "msm_phase_ms": {k: v / calls for k, v in tot["ms"].items()},
                         "int_utilisation": (adds / (acc_ms * 1e-3) / 1e9 / MIXED_ADD_PEAK_GOPS) if acc_ms else 0.0,
                         "kernel_ms_alone_on_gpu": alone_ms,
                         "frac_alone_on_gpu": (alg_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if alone_ms else None,
                         "step_algorithmic_bytes": step_bytes, "step_hbm_frac": step_bytes / (dt / max(1, timed_rows)) / 1e9 / HBM_PEAK_GBPS},
        }
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is timed at N = 1 only
            cores = usable_cores()
            ck_host = params.ck.download(0, info["primary_constraints"] + 8)
            ck2_host = ck2.download(0, max(info["secondary_wires"], info["secondary_constraints"]))
            sps, secs, n_cpu, cph = cpu_baseline(circuit, mine, z0, ck_host, args.cpu_seconds, cores, ivc=ivcs[0], ck2_host=ck2_host)
            out["cpu_baseline"] = {"value": sps, "unit": "steps/s", "cores": cores, "kind": "port",
                                   "sample": f"{n_cpu} folding steps with the CPU oracle (C++ restatement; not the Rust binary), like for like with a GPU step: per row the step circuit's witness "
                                             f"(rows of a batch side by side on the host threads), (A,B,C)·z, MSM(W), cross term, MSM(T) and folds of the primary instance, plus the two verifier "
                                             f"circuits' share — their (A,B,C)·z, four commitments, cross terms, folds and the hashes / scalar multiplications of their witness generation; "
                                             f"std::thread over the usable cores (affinity and cgroup quota), {secs:.1f} s",
                                   "seconds_per_step_by_phase": cph}
        print(json.dumps(out), flush=True)
    for v in ivcs:
        v.close()
    params.free()
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher's environment: start the N ranks (one process per GPU) the way the driver would —
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` — as a CHILD
    process, before anything here has touched a GPU; relay its one JSON line and exit with its code.  (The counterpart of
    /root/reference benchmark.sh:25-58, which starts and waits for its processes itself.)"""
    import socket
    import subprocess
    import torch
    if torch.cuda.device_count() < args.gpus and not args.share_gpus:      # (counting devices does not initialise the GPU)
        print(f"[bench] --gpus {args.gpus} but {torch.cuda.device_count()} device(s) visible; --share-gpus lets ranks share a device (not a scaling measurement)", file=sys.stderr)
        return 3
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ)
    if _vimz_lib.HW_QUEUES_DEFAULTED:
        env.pop("GPU_MAX_HW_QUEUES", None)      # every rank picks its own (8 with a GPU to itself, 4 when ranks share one)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=os.getcwd())
    got = False
    for line in proc.stdout:
        if line.startswith("{") and '"metric"' in line:
            got = True
        sys.stdout.write(line); sys.stdout.flush()
    rc = proc.wait()
    if rc == 0 and not got:
        print("[bench] the ranks exited without printing a result line", file=sys.stderr)
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--repeats", type=int, default=7, help="timed passes of exactly --steps rows each (ivc mode); `value` is the median pass")
    ap.add_argument("--warmup", type=int, default=150,
                    help="untimed rows proven first, the same way (default 150: proofs of more than 28 rows take the library's long-call "
                         "schedule - no host-evaluated head batch - and a 32-row warm-up left that path's first use inside the timed region)")
    ap.add_argument("--transformation", default="contrast")
    ap.add_argument("--resolution", default="HD")
    ap.add_argument("--batch", type=int, default=0, help="rows whose witnesses are generated together (0: folding.default_batch = 64: every BASELINE configuration below 64 GB of device memory, profiles/r05_batch_sweep.txt)")
    ap.add_argument("--segments", type=int, default=0, help="row segments folded concurrently on each GPU, own context + streams each, and merged into one proof (default in IVC mode: 3, 4 for 192 rows or more per GPU on eight or more host cores, 2 when the rank has fewer than six; 2 accumulators)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--msm-helpers", type=int, default=0, help="IVC mode: split every step's large MSM(T) by base range over this many helper contexts "
                    "(devices after this rank's, wrapping around; on a one-GPU box they share the device): one proof on several GPUs, SURVEY.md §8e")
    ap.add_argument("--no-compress", action="store_true", help="skip CompressedSNARK::prove / verify of the folded proof")
    ap.add_argument("--no-extras", action="store_true", help="skip the one-chain and Sonobe-backend extras of the default IVC run")
    ap.add_argument("--mode", default="ivc", choices=["ivc", "accumulator"])
    ap.add_argument("--host-head-batch", action="store_true", help="measure `value` under the library's default policy for short calls (first rows' Poseidon chains on a host pool) "
                                                                   "instead of the all-HIP schedule")
    ap.add_argument("--window-tables", type=int, default=15, help="window tables of the primary key in HBM (vimz_bases_precompute): 11 = per-window buckets, no host Horner; 13..16 = one shared bucket set; 0 = none")
    ap.add_argument("--proof-set", default="", help="comma-separated transformations: rank r proves proof_set[r %% len] (BASELINE config 5: independent proofs, replicas only)")
    ap.add_argument("--share-gpus", action="store_true", help="allow more ranks than visible GPUs (ranks r and r + n_devices share a device): evidence lines "
                    "on a one-GPU box, never a scaling claim; the line says so in `gpus_shared`")
    ap.add_argument("--cores", type=int, default=0, help="restrict this process (every rank: its own share) to this many host cores before anything starts — "
                    "what a rank has when eight ranks share a 16-core quota (VERDICT r3 #3); 0 = no restriction")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')} ranks: refusing to report a line for another N", file=sys.stderr)
        sys.exit(2)
    if args.cores > 0 and hasattr(os, "sched_setaffinity"):      # (before any thread exists: helper pools and the HIP runtime inherit the mask)
        cpus = sorted(os.sched_getaffinity(0))
        r = int(os.environ.get("LOCAL_RANK", "0"))
        mine = cpus[(r * args.cores) % len(cpus):][:args.cores] or cpus[:args.cores]
        os.sched_setaffinity(0, mine)
    rank = int(os.environ.get("RANK", "0"))
    if args.proof_set:
        ps = [t.strip() for t in args.proof_set.split(",") if t.strip()]
        args.transformation = ps[rank % len(ps)]
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        # (gloo's C++ side prints its connection banner on stdout: keep stdout for the ONE JSON line)
        sys.stdout.flush()
        keep = os.dup(1); os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo")      # timing barrier / max / final gather only: the data path has no collective
            dist.barrier()
        finally:
            sys.stdout.flush(); os.dup2(keep, 1); os.close(keep)
    ndev = torch.cuda.device_count()              # (counting devices does not initialise the GPU)
    if ndev < world and not args.share_gpus:
        print(f"[bench] --gpus {world} but {ndev} device(s) visible; --share-gpus lets ranks share a device (not a scaling measurement)", file=sys.stderr)
        sys.exit(3)
    args.gpus_shared = ndev < world
    device = local_rank % max(1, ndev)            # (--share-gpus: several ranks may share a GPU when the node has fewer GPUs than ranks)
    if torch.cuda.is_available():
        torch.cuda.set_device(device)

    from vimz_amd import folding, hip
    from vimz_amd.distributed import segment_bounds
    # three concurrent segments fill the GPU when the host has the cores to drive them (each: a fold thread, a launch-issuing thread, two
    # helpers); on two to five cores two segments do better — 865 against 541 steps/s on two cores, 636 as one chain (profiles/r04_cores.txt)
    # (IVC: three segments; FOUR for a proof of 192 rows or more per GPU on eight or more host cores — since the boolean-row form took 13 % of a step's instructions
    #  away a fourth chain finds room: 720 rows 1 312 -> 1 384 steps/s, 256 rows 1 272 -> 1 309, the 20-row window 871 -> 865; 27.6 instead of 21.1 GB at HD —
    #  profiles/r06_segments_sweep.txt)
    S = args.segments if args.segments > 0 else (((4 if args.steps >= 192 and usable_cores() >= 8 else 3) if usable_cores() >= 6 else 2) if args.mode == "ivc" else 2)
    # set-up with its parts side by side (folding.prepare_folding_overlapped): contexts and the step circuit together, then the keys, then — in IVC
    # mode — the segments' provers together; `setup_s` of the result line counts all of it
    t_setup = time.time()
    pre_ctxs = [hip.Context(device) for _ in range(S)] if os.environ.get("VIMZ_BENCH_SEQUENTIAL_CONTEXTS") else None      # (A/B: contexts one after another)
    ctxs, circuit, params, made_provers, setup_split = folding.prepare_folding_overlapped(device, S, args.transformation, args.resolution, window_tables=args.window_tables,
                                                                                         mode=("ivc" if args.mode == "ivc" and not args.msm_helpers and not os.environ.get("VIMZ_BENCH_SEQUENTIAL_PROVERS") else "none"),
                                                                                         batch=max(0, args.batch), ctxs=pre_ctxs)
    ctx = ctxs[0]
    args.setup_split = setup_split
    args.made_provers = made_provers if made_provers and made_provers[0] is not None else None
    if args.batch <= 0:
        args.batch = folding.default_batch(circuit)
    steps_all, z0 = build_inputs(args.transformation, args.resolution)
    n_rows = steps_all.shape[0]
    per_rank = args.warmup + (2 if args.mode == "ivc" else 1) * args.steps
    # global row list = concatenation of the ranks' segments; rank r folds rows [r*per_rank, (r+1)*per_rank) of it
    # (image rows are reused cyclically when the list is longer than the image)
    glob = [i % n_rows for i in range(world * per_rank)]
    lo, hi = segment_bounds(world * per_rank, world)[rank]
    mine = np.ascontiguousarray(steps_all[glob[lo:hi]])
    if args.mode == "ivc":
        return main_ivc(args, rank, world, dist, torch, ctxs, circuit, params, steps_all, glob, lo, hi, mine, z0, t_setup)
    provers = [hip.Prover(c, circuit, params.ck, max_batch=args.batch) for c in ctxs]   # the key and the shape are shared, read-only
    prover = provers[0]
    # IVC state at which this rank's segment starts (hash-only chain over the rows before it)
    z_start = _ints(prover.state_chain(z0, steps_all[glob[:lo]])[-1]) if lo else list(z0)
    setup_s = time.time() - t_setup

    from vimz_amd.distributed import fold_local_segments

    def sync_all():
        for c in ctxs:
            c.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # warm-up: the rank's first W rows, folded with the same local segmentation and merged, exactly like the timed part
    if args.warmup:
        fold_local_segments(provers, mine[:args.warmup], z_start)
        z_timed = _ints(prover.instance()["z"])
    else:
        z_timed = z_start
    warm_blob = prover.export() if args.warmup else None
    for c in ctxs:
        c.set_profiling(True)              # HIP events around every kernel of the MSM(T) launches on each context's stream
        c.msm_profile_totals(reset=True)
    sync_all()
    t0 = time.time()
    fold_local_segments(provers, mine[args.warmup:], z_timed)      # K rows: S concurrent segments + S-1 on-device merges
    sync_all()
    dt = time.time() - t0
    tots = [c.msm_profile_totals() for c in ctxs]
    tot = {"ms": {k: sum(t["ms"][k] for t in tots) for k in tots[0]["ms"]}, "calls": sum(t["calls"] for t in tots),
           "points": sum(t["points"] for t in tots), "entries": sum(t["entries"] for t in tots)}
    for c in ctxs:
        c.set_profiling(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    prof = prover.profile()
    if warm_blob is not None:              # put the warm-up rows back in front of the timed ones: one accumulator per rank
        timed_blob = prover.export()
        prover.reset(z_start)
        prover.merge(warm_blob)
        prover.merge(timed_blob)

    # host-side sequential final fold of the row segments (outside the timed region; reported separately)
    t_ff = time.time()
    if world > 1 and not args.proof_set:
        blob = prover.export()
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(np.asarray(blob).tobytes(), gathered, dst=0)
        if rank == 0:
            for r in range(1, world):
                prover.merge(np.frombuffer(gathered[r], dtype=np.uint8))
    final_fold_s = time.time() - t_ff
    ok = prover.verify() == 0 if (rank == 0 or args.proof_set) else True
    if args.proof_set and dist is not None:      # independent proofs: every rank verifies its own; AND them
        t = torch.tensor([1 if ok else 0], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(int(t[0]))
    inst = prover.instance()

    if rank == 0:
        nnz = circuit.nnz_a + circuit.nnz_b + circuit.nnz_c
        n_w, n_c = circuit.n_wires, circuit.n_constraints
        calls = max(1, tot["calls"])
        acc_ms = tot["ms"]["accumulate"] / calls                   # mean k_accum duration of the MSM(T) launches in the timed region
        msm_ms = sum(tot["ms"].values()) / calls
        alg_bytes = 96.0 * n_c                                     # 32 B scalar + 64 B affine base per point
        achieved = alg_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms else 0.0
        adds = tot["entries"] / calls
        traffic, traffic_src, traffic_stale = _pmc_traffic(f"{args.transformation}_step_{args.resolution}")
        step_bytes = 96 * n_w + 96 * n_c + 8 * nnz + 32 * n_w + 96 * n_c + 7 * 32 * n_c + 3 * 32 * n_w + 12 * 32 * n_c
        out = {
            "metric": "nova_folding_steps_per_sec",
            "value": world * args.steps / dt,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (256-bit Montgomery integers over BN254 Fr/Fq)",
            "data": "synthetic: rows of the reference sample image img2.png, contrast factor 1.4" + ("" if args.resolution == "HD" else f", upscaled to {args.resolution}"),
            "config": {"workload": f"{args.transformation}_step_{args.resolution}", "mode": "accumulator", "constraints": n_c, "wires": n_w, "nnz": nnz,
                       "rows_per_rank": args.steps, "segments_per_gpu": S, "witness_batch": args.batch, "parallelism": (f"{world} independent proofs ({args.proof_set}), replicas only" if args.proof_set else f"{world} independent row segments + host final fold")},
            "verified": bool(ok),
            "folded_steps_total": inst["steps"],
            "final_fold_s": final_fold_s if world > 1 else 0.0,
            "end_to_end_estimate_s": {"keygen_and_setup": setup_s, "fold_720_steps_one_gpu": 720 * dt / args.steps},
            "published_reference": {"contrast_HD_steps_per_s_cpu_server": 1.94, "source": "README.md:52 (720 steps / 371.7 s)"},
            "phase_ms_per_step": {k: 1e3 * v["seconds"] / max(1, (args.steps + args.warmup)) for k, v in prof.items()},
            "roofline": {"bound": "hbm", "kernel": "k_accum (bucket accumulation) of the MSM(T) launches in the timed region", "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": acc_ms, "launches": tot["calls"], "msm_gpu_ms": msm_ms,
                         "mixed_adds_per_launch": adds, "msm_phase_ms": {k: v / calls for k, v in tot["ms"].items()},
                         "int_utilisation": (adds / (acc_ms * 1e-3) / 1e9 / MIXED_ADD_PEAK_GOPS) if acc_ms else 0.0,
                         "step_algorithmic_bytes": step_bytes, "step_hbm_frac": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS},
        }
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is timed at N = 1 only
            cores = usable_cores()
            ck_host = params.ck.download(0, max(n_c, n_w))
            sps, secs, n_cpu, cph = cpu_baseline(circuit, mine, z_start, ck_host, args.cpu_seconds, cores)
            out["cpu_baseline"] = {"value": sps, "unit": "steps/s", "cores": cores, "kind": "port",
                                   "sample": f"{n_cpu} folding steps of the same workload with the CPU oracle (C++ restatement, std::thread over the usable cores (affinity and cgroup quota); not the Rust binary), {secs:.1f} s",
                                   "seconds_per_step_by_phase": cph}
        print(json.dumps(out), flush=True)
    for p_ in provers:
        p_.close()
    params.free()
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
