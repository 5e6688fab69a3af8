#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_distributed.py tests/test_gpu_field_msm.py -m gpu -x -q 2>&1 | tail -3
bash tools/refresh_profiles.sh r04 "bench pmc ranks cores" > $O/refresh.log 2>&1; tail -3 $O/refresh.log
