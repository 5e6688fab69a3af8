#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for ws in 16 8 4; do
  export VIMZ_TUNE="witness_sub=$ws"
  for rep in 1 2 3; do
  timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('witness_sub=$ws w256 3seg', round(d['value'],1), d['verified'], {k:round(v,3) for k,v in d['phase_ms_per_step_per_proof'].items() if 'host' in k or 'wait' in k})"
  done
  timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress --segments 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('witness_sub=$ws w256 1chain', round(d['value'],1), d['verified'])"
  timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('witness_sub=$ws w20', round(d['value'],1), d['verified'])"
  timeout 900 python bench.py --no-extras --no-cpu-baseline --no-compress --transformation contrast --resolution 4K --steps 96 --warmup 12 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('witness_sub=$ws 4K', round(d['value'],1), d['verified'])"
done
