#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout 300 python bench.py --no-extras --no-cpu-baseline --no-compress "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['verified'], d['config'].get('segments'))"; }
for c in 2 4; do
echo -n "cores $c default w256: "; b --cores $c
echo -n "cores $c default w256: "; b --cores $c
for h in 24 12 6; do echo -n "cores $c head $h w256: "; VIMZ_HEAD_ROWS=$h b --cores $c; done
echo -n "cores $c default w20: "; b --cores $c --steps 20 --warmup 5
echo -n "cores $c S=3 default w256: "; b --cores $c --segments 3
done
echo -n "cores 16 default w256: "; b
echo -n "cores 16 head 8 w256: "; VIMZ_HEAD_ROWS=8 b
echo -n "cores 16 head 12 w256: "; VIMZ_HEAD_ROWS=12 b
