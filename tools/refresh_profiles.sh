#!/bin/bash
# Regenerates the measurement set under gpurun_out/refresh/ on the GPU box (copy what is to be judged into profiles/):
#   the driver-style bench line and the long-window one, rocprofv3 kernel stats + the k_accum split of the same command, the
#   FETCH_SIZE / WRITE_SIZE counter passes (separate runs, counters only) summarised per kernel, whole-image runs and the bench
#   windows of the other configurations.   usage: tools/refresh_profiles.sh [round tag, default r02]
set -u
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
T="timeout 600"
$T python3 bench.py --steps 20 --warmup 5 > $O/${R}_bench_driver_window.json 2> $O/bench.err
$T python3 bench.py > $O/${R}_bench.json 2>> $O/bench.err
$T python3 bench.py --mode accumulator > $O/${R}_bench_accumulator.json 2>> $O/bench.err
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline --no-extras > $O/${R}_bench_under_rocprof.json 2> $O/rocprof.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${R}_bench_kernel_stats.csv
python3 tools/split_kernel_trace.py $(find $O/kt -name "*kernel_trace.csv" | head -1) k_accum > $O/${R}_k_accum_split.txt
python3 tools/trace_busy.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/${R}_trace_busy.txt 2>/dev/null
rm -rf $O/kt
for mode in ivc accumulator; do
  extra=""; [ $mode = accumulator ] && extra="--mode accumulator"
  $T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o pf -- python3 bench.py --no-cpu-baseline --no-extras --no-compress --steps 96 $extra > /dev/null 2>> $O/rocprof.err
  $T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -o pw -- python3 bench.py --no-cpu-baseline --no-extras --no-compress --steps 96 $extra > /dev/null 2>> $O/rocprof.err
  python3 tools/pmc_summary.py $(find $O/pf -name "*counter_collection.csv" | head -1) $(find $O/pw -name "*counter_collection.csv" | head -1) $O/${R}_pmc_summary_$mode.json > $O/${R}_pmc_summary_$mode.txt
  rm -rf $O/pf $O/pw
done
# whole images the way `vimz -b nova-snark -f <t>` sequences them
: > $O/${R}_e2e.jsonl
for cfg in "contrast HD 1 ivc" "grayscale HD 1 ivc" "blur HD 1 ivc" "crop HD 1 ivc" "contrast 4K 1 ivc" "resize 8K 1 ivc" "contrast HD 3 ivc" "contrast HD 2 accumulator"; do
  timeout 600 python3 tools/e2e.py $cfg 2>/dev/null | tail -1 >> $O/${R}_e2e.jsonl
done
# bench windows of the other BASELINE.json configurations
: > $O/${R}_configs.jsonl
run() { timeout 900 python3 bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 >> $O/${R}_configs.jsonl; }
run --transformation grayscale --resolution HD
run --transformation crop --resolution HD --steps 192
run --transformation contrast --resolution 4K --steps 192
run --transformation resize --resolution 8K --steps 192
run --transformation brightness --resolution 4K --steps 96
run --transformation sharpness --resolution 4K --steps 96
run --transformation blur --resolution 4K --steps 96
ls -la $O
