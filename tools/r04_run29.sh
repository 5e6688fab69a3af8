#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout 300 python bench.py --no-extras --no-cpu-baseline --no-compress "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), end=' ')"; }
VIMZ_DEBUG_TIMING=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --no-compress --cores 2 2>&1 | grep "head batch" | cut -c1-60 | sort | uniq -c
for h in default 24 12 8 6; do
echo -n "cores 2 head $h w256: "
for rep in 1 2 3 4; do if [ $h = default ]; then b --cores 2; else VIMZ_HEAD_ROWS=$h b --cores 2; fi; done; echo
done
