#!/bin/bash
# Regenerates the measurement set under gpurun_out/refresh/ on the GPU box (copy what is to be judged into profiles/):
#   bench lines (IVC default, accumulator), rocprofv3 kernel stats + the k_accum split of the same command, and the
#   FETCH_SIZE / WRITE_SIZE counter passes (separate runs, counters only) summarised per kernel, whole-image runs and the
#   bench windows of the other configurations.
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --mode accumulator > $O/bench_accumulator.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py > $O/bench_under_rocprof.json 2> $O/rocprof.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 tools/split_kernel_trace.py $(find $O/kt -name "*kernel_trace.csv" | head -1) k_accum > $O/k_accum_split.txt
rm -rf $O/kt
for mode in ivc accumulator; do
  extra=""; [ $mode = accumulator ] && extra="--mode accumulator"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o pf -- python3 bench.py --no-cpu-baseline --steps 96 $extra > /dev/null 2>> $O/rocprof.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -o pw -- python3 bench.py --no-cpu-baseline --steps 96 $extra > /dev/null 2>> $O/rocprof.err
  python3 tools/pmc_summary.py $(find $O/pf -name "*counter_collection.csv" | head -1) $(find $O/pw -name "*counter_collection.csv" | head -1) $O/pmc_summary_$mode.json
  rm -rf $O/pf $O/pw
done
# whole images the way `vimz -b nova-snark -f <t>` sequences them (-> profiles/r01_e2e.jsonl)
: > $O/e2e.jsonl
for cfg in "contrast HD 3 ivc" "grayscale HD 3 ivc" "blur HD 3 ivc" "crop HD 3 ivc" "contrast 4K 3 ivc" "resize 8K 3 ivc" "contrast HD 2 accumulator" "contrast HD 1 ivc"; do
  timeout 600 python3 tools/e2e.py $cfg 2>/dev/null | tail -1 >> $O/e2e.jsonl
done
# bench windows of the other BASELINE.json configurations (-> profiles/r01_configs.jsonl)
: > $O/configs.jsonl
run() { timeout 900 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> $O/configs.jsonl; }
run --transformation grayscale --resolution HD
run --transformation crop --resolution HD --steps 192
run --transformation contrast --resolution 4K --steps 192
run --transformation resize --resolution 8K --steps 192
run --transformation brightness --resolution 4K --steps 96
run --transformation sharpness --resolution 4K --steps 96
run --transformation blur --resolution 4K --steps 96
ls -la $O
