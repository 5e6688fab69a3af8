#!/bin/bash
# N > 1 evidence on the one-GPU box: bench lines with 2 and 4 ranks sharing the GPU (HIP IPC hand-over, tree of merges).
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" || exit 1
if [ "$1" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu.log
  tail -5 gpurun_out/r04/pytest_gpu.log
fi
timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r04/bench_plain_gpus2.json 2> gpurun_out/r04/bench_plain_gpus2.err; echo "plain --gpus 2 rc=$?"
for n in 2 4; do
  for rep in a b; do
  timeout 600 python bench.py --gpus $n --share-gpus --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r04/bench_${n}ranks_1gpu_w20_$rep.json 2> gpurun_out/r04/bench_${n}ranks_1gpu_w20_$rep.err; echo "n=$n w20 rc=$?"
  done
  timeout 900 python bench.py --gpus $n --share-gpus --steps 256 --warmup 32 --no-extras --no-cpu-baseline > gpurun_out/r04/bench_${n}ranks_1gpu_w256.json 2> gpurun_out/r04/bench_${n}ranks_1gpu_w256.err; echo "n=$n w256 rc=$?"
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r04/bench_1rank_w20.json 2> gpurun_out/r04/bench_1rank_w20.err; echo "n=1 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/bench_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, {k:d.get(k) for k in ("value","n_gpus","verified","state_chain_s","final_fold_s","final_fold_rank0_merging_s","prologue_s_max_over_ranks","sharding","fold_s","peak_device_bytes","peak_host_rss_bytes")})
    except Exception as e:
        print(f, "no line", e)
PY
for f in gpurun_out/r04/*.err; do echo "== $f"; tail -n 3 "$f"; done
