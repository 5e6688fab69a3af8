#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
bash tools/r04_nranks.sh > gpurun_out/r04/nranks.log 2>&1
tail -40 gpurun_out/r04/nranks.log | cut -c1-1500
timeout 600 python tools/msm_bench.py 305185 > gpurun_out/r04/msm_phases_tables.txt 2>&1
cat gpurun_out/r04/msm_phases_tables.txt
for wt in 0 16 15 14; do
  timeout 600 python bench.py --window-tables $wt --no-extras --no-cpu-baseline --no-compress > gpurun_out/r04/bench_w256_tables$wt.json 2> gpurun_out/r04/bench_w256_tables$wt.err
  timeout 600 python bench.py --window-tables $wt --no-extras --no-cpu-baseline --no-compress --steps 20 --warmup 5 > gpurun_out/r04/bench_w20_tables$wt.json 2> gpurun_out/r04/bench_w20_tables$wt.err
  timeout 600 python bench.py --window-tables $wt --no-extras --no-cpu-baseline --no-compress --segments 1 > gpurun_out/r04/bench_w256_1chain_tables$wt.json 2> gpurun_out/r04/bench_w256_1chain_tables$wt.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/bench_w*_tables*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        r=d["roofline"]
        print(f.split("/")[-1], "value %.1f"%d["value"], d["verified"], "k_accum %.3f ms"%r["kernel_ms"], {k:round(v,3) for k,v in r["msm_phase_ms"].items()}, {k:round(v,3) for k,v in d["phase_ms_per_step_per_proof"].items()})
    except Exception as e:
        print(f, "no line", e)
PY
