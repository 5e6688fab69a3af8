#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_field_msm.py -m gpu -x -q 2>&1 | tail -2
b() { timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['verified'])"; }
for rep in 1 2; do
echo -n "default            : "; b
echo -n "ones_dense=0       : "; VIMZ_TUNE=ones_dense=0 b
echo -n "dense_sub=24       : "; VIMZ_TUNE=dense_sub=24 b
echo -n "dense_sub=32       : "; VIMZ_TUNE=dense_sub=32 b
echo -n "dense_sub=12       : "; VIMZ_TUNE=dense_sub=12 b
done
echo -n "w20 default        : "; b --steps 20 --warmup 5
echo -n "w20 dense_sub=32   : "; VIMZ_TUNE=dense_sub=32 b --steps 20 --warmup 5
echo -n "1chain default     : "; b --segments 1
echo -n "1chain dense_sub=32: "; VIMZ_TUNE=dense_sub=32 b --segments 1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $O/pv -o pv -- python3 tools/msm_bench.py 305185 > $O/msm_bench.txt 2>> $O/rocprof.err
C=$(find $O/pv -name "*counter_collection.csv" | head -1)
python3 tools/valu_budget.py $C 1 SQ_INSTS_VALU split > $O/valu_msm_bench.txt; tail -25 $O/valu_msm_bench.txt; grep -E "entries" $O/msm_bench.txt | cut -c1-150
rm -rf $O/pv
