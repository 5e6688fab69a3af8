#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout 300 python bench.py --no-extras --no-cpu-baseline --no-compress "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['verified'], d['config'].get('segments'))"; }
for c in 2 4; do for h in 24 8 6 4 3 2; do
echo -n "cores $c head $h w20: "; VIMZ_HEAD_ROWS=$h b --cores $c --steps 20 --warmup 5
done; done
for h in 24 8 4; do
echo -n "cores 2 head $h w256: "; VIMZ_HEAD_ROWS=$h b --cores 2
done
for h in 24 12 8 5; do
echo -n "cores 16 head $h w20: "; VIMZ_HEAD_ROWS=$h b --steps 20 --warmup 5
done
