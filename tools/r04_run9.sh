#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04b; mkdir -p $O
timeout 300 ./tools/ubench > $O/r04_ubench_mfma_bound.txt 2>&1; tail -12 $O/r04_ubench_mfma_bound.txt
timeout 900 python bench.py > $O/r04_bench.json 2> $O/bench.err; echo "default rc=$?"
for rep in a b c; do timeout 600 python bench.py --steps 20 --warmup 5 --no-extras > $O/r04_bench_driver_window_$rep.json 2>> $O/bench.err; done
timeout 900 python bench.py --cores 2 --no-extras --no-cpu-baseline > $O/r04_bench_2cores.json 2>> $O/bench.err; echo "cores2 rc=$?"
timeout 900 python bench.py --cores 2 --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $O/r04_bench_2cores_driver_window.json 2>> $O/bench.err
timeout 900 python bench.py --gpus 2 --share-gpus --no-extras --no-cpu-baseline > $O/r04_bench_2ranks_on_1gpu.json 2> $O/ranks.err; echo "2ranks rc=$?"
timeout 900 python bench.py --gpus 2 --share-gpus --cores 2 --no-extras --no-cpu-baseline > $O/r04_bench_2ranks_on_1gpu_2cores_each.json 2>> $O/ranks.err; echo "2ranks 2cores rc=$?"
timeout 1200 python bench.py --gpus 2 --share-gpus --transformation resize --resolution 8K --steps 64 --warmup 8 --no-extras --no-cpu-baseline --no-compress > $O/r04_bench_2ranks_on_1gpu_8K.json 2>> $O/ranks.err; echo "8K 2ranks rc=$?"
timeout 1200 python bench.py --gpus 4 --share-gpus --transformation resize --resolution 8K --steps 32 --warmup 8 --segments 2 --batch 32 --no-extras --no-cpu-baseline --no-compress > $O/r04_bench_4ranks_on_1gpu_8K.json 2>> $O/ranks.err; echo "8K 4ranks rc=$?"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04b/r04_bench*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        sh=d.get("sharding") or {}
        print(f.split("/")[-1], "value %.1f"%d["value"], "n",d["n_gpus"], "ver",d["verified"], "cores", d.get("host_cores_per_rank"), "prolog %.4f"%d["prologue_s_max_over_ranks"], "final %.4f"%d["final_fold_s"], "fold %.4f"%d["fold_s"], [(h["from"], round(h["open_s"],4), round(h["merge_s"],4)) for h in (sh.get("hand_overs_to_rank0") or [])], {k:round(v,3) for k,v in d["phase_ms_per_step_per_proof"].items() if "host" in k}, "1chain", (d.get("one_chain") or {}).get("steps_per_s"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "mem", d.get("peak_device_bytes"), d.get("peak_host_rss_bytes"))
        if d.get("cpu_baseline"): print("   cpu:", d["cpu_baseline"]["seconds_per_step_by_phase"], d["cpu_baseline"]["sample"][:80])
    except Exception as e:
        print(f, "no line", e)
PY
tail -5 $O/bench.err $O/ranks.err | cut -c1-300
