#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
export VIMZ_DEBUG_TIMING=1
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress > gpurun_out/r04/t_w256.json 2> gpurun_out/r04/t_w256.err
grep "timing" gpurun_out/r04/t_w256.err | tail -12
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress --steps 20 --warmup 5 > gpurun_out/r04/t_w20.json 2> gpurun_out/r04/t_w20.err
grep "timing" gpurun_out/r04/t_w20.err | tail -12
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress --segments 1 > gpurun_out/r04/t_1c.json 2> gpurun_out/r04/t_1c.err
grep "timing" gpurun_out/r04/t_1c.err | tail -4
python - <<'PY'
import json
for t in ("w256","w20","1c"):
    d=json.loads([l for l in open(f"gpurun_out/r04/t_{t}.json") if l.startswith("{")][-1])
    print(t, round(d["value"],1), {k:round(v,3) for k,v in d["phase_ms_per_step_per_proof"].items()})
PY
