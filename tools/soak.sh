#!/bin/bash
# Soak of the final code of a round on one GPU box: repeated merged proofs on the same provers (times flat, device memory flat), generations of
# provers on the same contexts, the MSM stress tools, whole images; every proof verified by the tools.  usage: tools/soak.sh > profiles/rNN_soak.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
echo "== stress_merge: 60 merged proofs of 48 rows, then 20 of 150 rows (the long-call schedule: no head batch)"
timeout 900 python3 tools/stress_merge.py 60 48 2>&1 | tail -4 | cut -c1-400
timeout 900 python3 tools/stress_merge.py 20 150 2>&1 | tail -4 | cut -c1-400
echo "== stress_merge with every row's witness on the GPU (VIMZ_HEAD_ROWS=0: the segments start together, deferred start states): 60 proofs of 48 rows, 40 of 21 rows"
VIMZ_HEAD_ROWS=0 timeout 900 python3 tools/stress_merge.py 60 48 2>&1 | tail -4 | cut -c1-400
VIMZ_HEAD_ROWS=0 timeout 900 python3 tools/stress_merge.py 40 21 2>&1 | tail -4 | cut -c1-400
echo "== stress_large_msm (the large MSM's tails: split heavy buckets, table path against the plain one), default reduce and bit planes"
timeout 900 python3 tools/stress_large_msm.py 305185 20 3 2>&1 | tail -1 | cut -c1-300
VIMZ_TUNE=reduce_planes=1 timeout 900 python3 tools/stress_large_msm.py 305185 20 3 2>&1 | tail -1 | cut -c1-300
echo "== prover_generations nova2cf"
timeout 900 python3 tools/prover_generations.py nova2cf 2>&1 | grep -E "steps/s|verified|generation" | tail -8 | cut -c1-300
echo "== stress_small_msm / stress_cyclefold"
timeout 600 python3 tools/stress_small_msm.py 2>&1 | tail -2 | cut -c1-300
timeout 900 python3 tools/stress_cyclefold.py 2>&1 | tail -3 | cut -c1-400
echo "== decider: the Sonobe path end to end at contrast HD, twice (fold, decider proof, local verification of the 25 words)"
for rep in 1 2; do timeout 600 python3 tools/e2e.py contrast HD 1 cyclefold 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cyclefold + decider', d['steps'], round(d['steps_per_s'],1), d['verified'], 'decider verified', d['decider']['verified'], 'prove', round(d['decider']['prove_s']['total'],3), 'wall', round(d['wall_total_s'],2))"; done
echo "== whole images"
for cfg in "contrast HD 3 ivc" "contrast HD 3 ivc" "contrast HD 1 ivc" "crop HD 3 ivc" "contrast HD 2 accumulator" "contrast HD 3 cyclefold"; do timeout 600 python3 tools/e2e.py $cfg 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('whole image', d['config'], d['mode'], d['segments'], d['steps'], round(d['steps_per_s'],1), d['verified'], d['final_state'][0][:18])"; done
echo "== bench, three times"
for rep in 1 2 3; do timeout 600 python3 bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value'],1), d['verified'], 'compressed', d['compressed_snark']['verified'] if d.get('compressed_snark') else None)"; done
