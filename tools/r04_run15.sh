#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04f; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline --no-extras --no-compress > $O/bench.json 2> $O/err.txt
T=$(find $O/kt -name "*kernel_trace.csv" | head -1)
head -1 $T
python3 tools/msm_chain_gaps.py $T > $O/r04_msm_chain_gaps_HD.txt; cat $O/r04_msm_chain_gaps_HD.txt
rm -rf $O/kt
