#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_decider.py -m gpu -x -q > $O/pytest_decider.log 2>&1; tail -25 $O/pytest_decider.log | cut -c1-300
timeout 900 python bench.py --cores 2 --no-extras --no-cpu-baseline > $O/r04_bench_2cores.json 2> $O/bench.err; echo "cores2 rc=$?"
timeout 900 python bench.py --cores 4 --no-extras --no-cpu-baseline > $O/r04_bench_4cores.json 2>> $O/bench.err; echo "cores4 rc=$?"
timeout 900 python bench.py --gpus 2 --share-gpus --cores 2 --no-extras --no-cpu-baseline > $O/r04_bench_2ranks_on_1gpu_2cores_each.json 2>> $O/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04c/r04_bench*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "value %.1f"%d["value"], "cores", d.get("host_cores_per_rank"), {k:round(v,3) for k,v in d["phase_ms_per_step_per_proof"].items()})
    except Exception as e:
        print(f, "no line", e)
PY
