#!/usr/bin/env python3
"""A/B runs of bench.py under environment switches, one summary line each (GPU box).
usage: ab_bench.py [--reps N] [--args "<bench.py flags>"] VAR=a,b,c [VAR2=x,y ...]   (cartesian product; "-" = unset)"""
import itertools
import json
import os
import subprocess
import sys

reps, extra, axes = 2, "--no-extras --no-compress --no-cpu-baseline", []
a = sys.argv[1:]
while a:
    x = a.pop(0)
    if x == "--reps":
        reps = int(a.pop(0))
    elif x == "--args":
        extra = a.pop(0)
    else:
        k, v = x.split("=", 1)
        axes.append((k, v.split(",")))
root = __file__.rsplit("/tools/", 1)[0]
for combo in itertools.product(*[v for _, v in axes]):
    env = dict(os.environ)
    for (k, _), val in zip(axes, combo):
        env.pop(k, None)
        if val != "-":
            env[k] = val
    for _ in range(reps):
        r = subprocess.run([sys.executable, root + "/bench.py"] + extra.split(), env=env, capture_output=True, text=True, timeout=600)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            print(combo, "FAILED", r.stderr[-300:])
            continue
        p = d.get("phase_ms_per_step_per_proof", {})
        ph = d["roofline"].get("msm_phase_ms") or {}
        print(" ".join(f"{k}={v}" for (k, _), v in zip(axes, combo)), round(d["value"], 1), d["verified"],
              "sec", round(p.get("wait_secondary_msm", 0), 3), "pri", round(p.get("wait_primary_msm", 0), 3),
              "c1", round(p.get("verifier_circuit_primary_host", 0), 3), "c2", round(p.get("verifier_circuit_secondary_host", 0), 3),
              {k: round(v, 3) for k, v in ph.items()}, flush=True)
