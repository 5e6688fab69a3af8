#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
export VIMZ_BENCH_DEBUG=1
run() { tag=$1; shift; timeout 600 "$@" > gpurun_out/r04/x_$tag.json 2> gpurun_out/r04/x_$tag.err; python - "$tag" <<'PY'
import json,sys
tag=sys.argv[1]
try:
    d=json.loads([l for l in open(f"gpurun_out/r04/x_{tag}.json") if l.startswith("{")][-1])
    sh=d.get("sharding") or {}
    print(tag, "value %.1f"%d["value"], "prolog %.4f"%d["prologue_s_max_over_ranks"], "final %.4f"%d["final_fold_s"], "fold %.4f"%d["fold_s"], "allg %.4f"%sh.get("allgather_s_max",0), [(h["from"], round(h["open_s"],4), round(h["merge_s"],4)) for h in (sh.get("hand_overs_to_rank0") or [])])
except Exception as e:
    print(tag, "no line", e)
PY
}
B="python bench.py --gpus 4 --share-gpus --steps 20 --warmup 5 --no-extras --no-cpu-baseline"
for rep in 1 2 3; do run q4_$rep $B; done
for rep in 1 2 3; do GPU_MAX_HW_QUEUES=2 run q2_$rep $B; done
for rep in 1 2; do run seg1_$rep $B --segments 1; done
for rep in 1 2; do run seg2_$rep $B --segments 2; done
for rep in 1 2; do GPU_MAX_HW_QUEUES=2 run seg2q2_$rep $B --segments 2; done
