#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04k; mkdir -p $O
for od in 1 0; do
VIMZ_TUNE=ones_dense=$od timeout 900 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $O/pv$od -o pv -- python3 bench.py --no-cpu-baseline --no-extras --no-compress --steps 96 > /dev/null 2>> $O/rocprof.err
C=$(find $O/pv$od -name "*counter_collection.csv" | head -1)
python3 tools/valu_budget.py $C 224 SQ_INSTS_VALU split > $O/valu_ones_dense_$od.txt; grep -E "per step|k_ones" $O/valu_ones_dense_$od.txt
rm -rf $O/pv$od
done
timeout 900 python -m pytest tests/test_gpu_field_msm.py tests/test_gpu_ivc.py -m gpu -x -q 2>&1 | tail -2
