#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout 300 python bench.py --no-extras --no-cpu-baseline --no-compress "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['verified'], {k:round(v,3) for k,v in d['phase_ms_per_step_per_proof'].items() if k!='secondary_gpu'})"; }
for rep in 1 2; do
echo -n "default      : "; b
echo -n "lookahead    : "; VIMZ_IVC_LOOKAHEAD=1 b
done
echo -n "S=4 lookahead    : "; VIMZ_IVC_LOOKAHEAD=1 b --segments 4
echo -n "S=2 lookahead    : "; VIMZ_IVC_LOOKAHEAD=1 b --segments 2
echo -n "1chain default   : "; b --segments 1
echo -n "1chain lookahead : "; VIMZ_IVC_LOOKAHEAD=1 b --segments 1
echo -n "w20 default      : "; b --steps 20 --warmup 5
echo -n "w20 lookahead    : "; VIMZ_IVC_LOOKAHEAD=1 b --steps 20 --warmup 5
echo -n "4K default       : "; b --transformation contrast --resolution 4K --steps 96 --warmup 12
echo -n "4K lookahead     : "; VIMZ_IVC_LOOKAHEAD=1 b --transformation contrast --resolution 4K --steps 96 --warmup 12
VIMZ_IVC_LOOKAHEAD=1 VIMZ_DEBUG_TIMING=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --no-compress 2>&1 | grep -E "wait_primary" | tail -3 | cut -c1-400
