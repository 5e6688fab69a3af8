#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 300 python -m pytest tests/test_gpu_field_msm.py -m gpu -x -q 2>&1 | tail -2
for lean in 0 1 2; do
  export VIMZ_TUNE="small_lean=$lean"
  echo "== small_lean=$lean"
  timeout 120 python tools/small_msm_bench.py plain
  for rep in 1 2; do
  timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w256 3seg', round(d['value'],1), d['verified'])"
  done
  timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress --segments 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w256 1chain', round(d['value'],1), d['verified'])"
  for rep in 1 2; do
  timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w20 3seg', round(d['value'],1), d['verified'])"
  done
done
