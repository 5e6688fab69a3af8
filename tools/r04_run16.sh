#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_field_msm.py tests/test_gpu_fold.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/msm_bench.py 305185 2>&1 | grep -E "dense    c=11|tables c=15|witness  c=11"
timeout 300 python tools/stress_small_msm.py 2>&1 | tail -3
for rep in 1 2 3; do
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w256 3seg', round(d['value'],1), d['verified'], {k:round(v,3) for k,v in d['roofline']['msm_phase_ms'].items()}, {k:round(v,3) for k,v in d['phase_ms_per_step_per_proof'].items() if 'wait' in k})"
done
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress --segments 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w256 1chain', round(d['value'],1), d['verified'])"
for rep in 1 2 3; do
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-compress --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w20', round(d['value'],1), d['verified'])"
done
timeout 900 python bench.py --no-extras --no-cpu-baseline --no-compress --transformation contrast --resolution 4K --steps 96 --warmup 12 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4K', round(d['value'],1), d['verified'])"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline --no-extras --no-compress > $O/bench.json 2> $O/err.txt
T=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 tools/msm_chain_gaps.py $T 340736 > $O/r04_msm_chain_gaps_HD.txt; cat $O/r04_msm_chain_gaps_HD.txt
rm -rf $O/kt
