#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python tools/e2e.py hash HD 1 cyclefold > $O/e2e_cf_hash.json 2> $O/e2e.err; tail -c 1800 $O/e2e_cf_hash.json; echo
timeout 1500 python tools/e2e.py contrast HD 1 cyclefold > $O/e2e_cf_contrast.json 2>> $O/e2e.err; tail -c 1800 $O/e2e_cf_contrast.json; echo
tail -5 $O/e2e.err | cut -c1-300
for seg in 1 2 3; do
timeout 900 python bench.py --cores 2 --segments $seg --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cores2 segments $seg', round(d['value'],1), d['verified'])"
done
