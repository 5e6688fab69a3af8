#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04i; mkdir -p $O
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $O/pv -o pv -- python3 bench.py --no-cpu-baseline --no-extras --no-compress --steps 96 > /dev/null 2>> $O/rocprof.err
C=$(find $O/pv -name "*counter_collection.csv" | head -1)
head -2 $C
python3 tools/valu_budget.py $C 224 SQ_INSTS_VALU split > $O/valu_split.txt; cat $O/valu_split.txt
rm -rf $O/pv
