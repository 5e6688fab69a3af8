#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for S in 4 5 6; do
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress --segments $S 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w256 S=$S', round(d['value'],1), d['verified'])"
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress --segments $S --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w20 S=$S', round(d['value'],1), d['verified'])"
done
bash tools/refresh_profiles.sh r04 "bench rocprof pmc valu" > $O/refresh.log 2>&1; tail -5 $O/refresh.log
