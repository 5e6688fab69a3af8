#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
VIMZ_DEBUG_TIMING=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --no-compress 2>&1 | grep -E "wait_primary|fold of" | tail -8 | cut -c1-400
echo ---- one chain
VIMZ_DEBUG_TIMING=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --no-compress --segments 1 2>&1 | grep -E "wait_primary|fold of" | tail -3 | cut -c1-400
