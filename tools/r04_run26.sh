#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
for c in 2 4 16; do
echo "== cores $c, 20-row window"
VIMZ_DEBUG_TIMING=1 timeout 300 python bench.py --cores $c --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-compress 2>&1 | grep -E "timing|metric" | tail -14 | cut -c1-330 | sed 's/"unit".*"state_chain_s"/ ... "state_chain_s"/' | cut -c1-420
done
