#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/refresh_profiles.sh r04 all > gpurun_out/refresh_all.log 2>&1; tail -2 gpurun_out/refresh_all.log
