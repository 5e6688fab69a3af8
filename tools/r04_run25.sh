#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout 300 python bench.py --no-extras --no-cpu-baseline --no-compress "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['verified'], {k:round(v,3) for k,v in d['phase_ms_per_step_per_proof'].items() if k!='secondary_gpu'})"; }
for rep in 1 2; do
echo -n "default      : "; b
echo -n "s3 hi        : "; VIMZ_X_S3HI=1 b
done
echo -n "w20 default      : "; b --steps 20 --warmup 5
echo -n "w20 s3 hi        : "; VIMZ_X_S3HI=1 b --steps 20 --warmup 5
echo -n "1chain s3 hi     : "; VIMZ_X_S3HI=1 b --segments 1
