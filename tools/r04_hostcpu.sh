#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout 300 python bench.py --no-extras --no-cpu-baseline --no-compress "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_cpu']; print(round(d['value'],1), d['verified'], 'cores busy', round(h['cores_busy_rank0'],2), 'core-ms/step', round(h['core_ms_per_step_rank0'],3))"; }
for rep in 1 2 3; do
echo -n "2 cores S=2 w256 poll   : "; b --cores 2
echo -n "2 cores S=2 w256 hipsync: "; VIMZ_X_POLL=0 b --cores 2
done
echo -n "2 cores S=3 w256 poll   : "; b --cores 2 --segments 3
echo -n "4 cores S=2 w256 poll   : "; b --cores 4
echo -n "4 cores S=2 w256 hipsync: "; VIMZ_X_POLL=0 b --cores 4
echo -n "4 cores S=3 w256 poll   : "; b --cores 4 --segments 3
echo -n "16 cores S=3 w256 hipsync: "; b
echo -n "16 cores S=3 w256 poll   : "; VIMZ_X_POLL=1 b
echo -n "2 cores w20 poll   : "; b --cores 2 --steps 20 --warmup 5
echo -n "2 cores w20 hipsync: "; VIMZ_X_POLL=0 b --cores 2 --steps 20 --warmup 5
