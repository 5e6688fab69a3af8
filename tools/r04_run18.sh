#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
for Q in 8 12 16 24; do for S in 3 4; do
GPU_MAX_HW_QUEUES=$Q timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress --segments $S 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w256 Q=$Q S=$S', round(d['value'],1), d['verified'], {k:round(v,3) for k,v in d['phase_ms_per_step_per_proof'].items()})"
done; done
