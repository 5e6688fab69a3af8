#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
export VIMZ_BENCH_DEBUG=1
for rep in a b c; do
  timeout 600 python bench.py --gpus 4 --share-gpus --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r04/bench_4ranks_1gpu_w20_$rep.json 2> gpurun_out/r04/bench_4ranks_1gpu_w20_$rep.err; echo "n=4 w20 rc=$?"
  grep "bench rank" gpurun_out/r04/bench_4ranks_1gpu_w20_$rep.err | cut -c1-700
done
for rep in a b; do
  timeout 600 python bench.py --gpus 2 --share-gpus --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r04/bench_2ranks_1gpu_w20_$rep.json 2> gpurun_out/r04/bench_2ranks_1gpu_w20_$rep.err; echo "n=2 w20 rc=$?"
done
timeout 900 python bench.py --gpus 4 --share-gpus --steps 256 --warmup 32 --no-extras --no-cpu-baseline > gpurun_out/r04/bench_4ranks_1gpu_w256.json 2> gpurun_out/r04/bench_4ranks_1gpu_w256.err
timeout 900 python bench.py --gpus 2 --share-gpus --steps 256 --warmup 32 --no-extras --no-cpu-baseline > gpurun_out/r04/bench_2ranks_1gpu_w256.json 2> gpurun_out/r04/bench_2ranks_1gpu_w256.err
unset VIMZ_BENCH_DEBUG
for rep in a b c; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r04/bench_1rank_w20_$rep.json 2> gpurun_out/r04/bench_1rank_w20.err; echo "n=1 rc=$?"
done
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
timeout 900 python -m pytest tests/test_gpu_merge.py tests/test_gpu_fold.py tests/test_distributed.py -m gpu -x -q 2>&1 | tail -3
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/bench_*ranks*.json"))+sorted(glob.glob("gpurun_out/r04/bench_1rank_w20_*.json"))+["gpurun_out/r04/bench_default.json"]:
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        sh=d.get("sharding") or {}
        print(f.split("/")[-1], "value %.1f"%d["value"], "ver",d["verified"], "chain %.4f"%d["state_chain_s"], "prolog %.4f"%d["prologue_s_max_over_ranks"], "final %.4f"%d["final_fold_s"], "r0merge %.4f"%d["final_fold_rank0_merging_s"], "fold %.4f"%d["fold_s"], {k:(round(v,4) if isinstance(v,float) else v) for k,v in sh.items() if k!="hand_overs_to_rank0"}, [ {k:(round(v,4) if isinstance(v,float) else v) for k,v in h.items()} for h in (sh.get("hand_overs_to_rank0") or [])], "setup", round(d["end_to_end_estimate_s"]["keygen_and_setup"],3), d.get("one_chain"))
    except Exception as e:
        print(f, "no line", e)
PY
