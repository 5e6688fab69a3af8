#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout 300 python bench.py --no-extras --no-cpu-baseline --no-compress "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['verified'], {k:round(v,3) for k,v in d['phase_ms_per_step_per_proof'].items() if k!='secondary_gpu' and k!='producer_wait'})"; }
echo -n "default      : "; b
for sp in 32 64 32s 64s 96s 128s; do
echo -n "split $sp     : "; VIMZ_CU_SPLIT=$sp b
done
echo -n "default      : "; b
echo -n "1chain default : "; b --segments 1
echo -n "1chain 64s     : "; VIMZ_CU_SPLIT=64s b --segments 1
echo -n "w20 default    : "; b --steps 20 --warmup 5
echo -n "w20 64s        : "; VIMZ_CU_SPLIT=64s b --steps 20 --warmup 5
