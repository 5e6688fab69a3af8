#!/bin/bash
# Regenerates the measurement set under gpurun_out/refresh/ on the GPU box (copy what is to be judged into profiles/):
#   the driver-style bench line and the long-window one, rocprofv3 kernel stats + the k_accum split of the same command, the
#   FETCH_SIZE / WRITE_SIZE counter passes (separate runs, counters only) summarised per kernel — for the HD headline and for the
#   4K / 8K shapes —, whole-image runs, the bench windows of the other configurations, the N > 1 lines on the one GPU of the box.
#   usage: tools/refresh_profiles.sh [round tag, default r06] [parts: all | bench | rocprof | pmc | valu | e2e | configs | ranks | cores | cyclefold | round5 | round6 | stalls]
set -u
R=${1:-r06}
PARTS=${2:-all}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; mkdir -p $O
T="timeout 900"
want() { [ "$PARTS" = all ] || echo "$PARTS" | grep -q "$1"; }
if want bench; then
  $T python3 bench.py --steps 20 --warmup 5 > $O/${R}_bench_driver_window.json 2> $O/bench.err
  for rep in b c d e; do $T python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/${R}_bench_driver_window_$rep.json 2>> $O/bench.err; done
  $T python3 tools/msm_bench.py 305185 > $O/${R}_msm_phases_tables.txt 2>&1
  $T python3 bench.py > $O/${R}_bench.json 2>> $O/bench.err
  $T python3 bench.py --segments 1 --no-cpu-baseline > $O/${R}_bench_one_chain.json 2>> $O/bench.err
  $T python3 bench.py --mode accumulator --no-cpu-baseline > $O/${R}_bench_accumulator.json 2>> $O/bench.err
  (python3 tools/small_msm_bench.py plain; python3 tools/small_msm_bench.py mult) > $O/${R}_small_msm.txt 2>/dev/null
fi
kstats() {   # kstats <tag> <bench args...>: rocprofv3 kernel trace + stats of one bench command
  local tag=$1; shift
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline --no-extras --no-compress "$@" > $O/${R}_bench_under_rocprof_$tag.json 2> $O/rocprof.err
  cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${R}_kernel_stats_$tag.csv
  python3 tools/split_kernel_trace.py $(find $O/kt -name "*kernel_trace.csv" | head -1) k_accum > $O/${R}_k_accum_split_$tag.txt
  python3 tools/split_kernel_trace.py $(find $O/kt -name "*kernel_trace.csv" | head -1) k_msm_small > $O/${R}_k_msm_small_split_$tag.txt
  python3 tools/trace_busy.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/${R}_trace_busy_$tag.txt 2>/dev/null
  rm -rf $O/kt
}
if want rocprof; then
  kstats HD
  kstats HD_one_chain --segments 1
  kstats 4K --transformation contrast --resolution 4K --steps 96 --warmup 12
  kstats 8K --transformation resize --resolution 8K --steps 96 --warmup 12
fi
# (counter passes: one timed pass, and contexts / provers created one after another on the main thread — with the set-up's short-lived threads issuing HIP
#  calls rocprofv3's counter collection aborted on a dangling correlation id and hung, round 5)
export_seq() { export VIMZ_BENCH_SEQUENTIAL_CONTEXTS=1 VIMZ_BENCH_SEQUENTIAL_PROVERS=1; }
unset_seq() { unset VIMZ_BENCH_SEQUENTIAL_CONTEXTS VIMZ_BENCH_SEQUENTIAL_PROVERS; }
pmc() {      # pmc <tag> <bench args...>: FETCH_SIZE and WRITE_SIZE in separate passes (counters only)
  local tag=$1; shift
  export_seq
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o pf -- python3 bench.py --no-cpu-baseline --no-extras --no-compress --repeats 1 "$@" > /dev/null 2>> $O/rocprof.err
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -o pw -- python3 bench.py --no-cpu-baseline --no-extras --no-compress --repeats 1 "$@" > /dev/null 2>> $O/rocprof.err
  unset_seq
  python3 tools/pmc_summary.py $(find $O/pf -name "*counter_collection.csv" | head -1) $(find $O/pw -name "*counter_collection.csv" | head -1) $O/${R}_pmc_summary_$tag.json > $O/${R}_pmc_summary_$tag.txt
  rm -rf $O/pf $O/pw
}
if want pmc; then
  pmc ivc --steps 96
  pmc accumulator --steps 96 --mode accumulator
  pmc ivc_4K --transformation contrast --resolution 4K --steps 48 --warmup 12
  pmc ivc_8K --transformation resize --resolution 8K --steps 48 --warmup 12
fi
if want valu; then     # vector instructions per step and kernel (the instruction budget of DESIGN.md §8)
  export_seq
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $O/pv -o pv -- python3 bench.py --no-cpu-baseline --no-extras --no-compress --repeats 1 --steps 96 --warmup 32 > /dev/null 2>> $O/rocprof.err
  unset_seq
  python3 tools/valu_budget.py $(find $O/pv -name "*counter_collection.csv" | head -1) 224 SQ_INSTS_VALU split > $O/${R}_valu_budget.txt
  rm -rf $O/pv
fi
if want e2e; then      # whole images the way `vimz -b nova-snark -f <t>` sequences them
  : > $O/${R}_e2e.jsonl
  for cfg in "contrast HD 4 ivc" "contrast HD 3 ivc" "contrast HD 1 ivc" "grayscale HD 4 ivc" "blur HD 4 ivc" "crop HD 4 ivc" "crop HD 3 ivc" "contrast 4K 4 ivc" "contrast 4K 3 ivc" "contrast 4K 1 ivc" "resize 8K 4 ivc" "resize 8K 3 ivc" "resize 8K 1 ivc" "contrast HD 2 accumulator"; do
    timeout 900 python3 tools/e2e.py $cfg 2>/dev/null | tail -1 >> $O/${R}_e2e.jsonl
  done
fi
if want configs; then  # bench windows of the other BASELINE.json configurations
  : > $O/${R}_configs.jsonl
  run() { timeout 900 python3 bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 >> $O/${R}_configs.jsonl; }
  run --transformation grayscale --resolution HD
  run --transformation crop --resolution HD --steps 192
  run --transformation contrast --resolution 4K --steps 192
  run --transformation resize --resolution 8K --steps 192
  run --transformation brightness --resolution 4K --steps 96
  run --transformation sharpness --resolution 4K --steps 96
  run --transformation blur --resolution 4K --steps 96
fi
if want ranks; then    # N > 1 on the one GPU of the box (--share-gpus: evidence of the sharded path, not of scaling): ONE object out of the ranks' segments,
                       # HIP IPC hand-over up a tree; the proof set of BASELINE config 5; the 8K shapes; ranks on two host cores each
  B="timeout 1200 python3 bench.py --share-gpus --no-extras --no-cpu-baseline"
  $B --gpus 2 > $O/${R}_bench_2ranks_on_1gpu_ivc.json 2> $O/ranks.err
  for rep in a b c; do $B --gpus 2 --steps 20 --warmup 5 > $O/${R}_bench_2ranks_on_1gpu_driver_window_$rep.json 2>> $O/ranks.err; done
  for rep in a b c; do $B --gpus 4 --steps 20 --warmup 5 > $O/${R}_bench_4ranks_on_1gpu_driver_window_$rep.json 2>> $O/ranks.err; done
  $B --gpus 4 > $O/${R}_bench_4ranks_on_1gpu_ivc.json 2>> $O/ranks.err
  $B --gpus 2 --cores 2 > $O/${R}_bench_2ranks_on_1gpu_2cores_each.json 2>> $O/ranks.err
  $B --gpus 2 --transformation resize --resolution 8K --steps 64 --warmup 8 --no-compress > $O/${R}_bench_2ranks_on_1gpu_8K.json 2>> $O/ranks.err
  $B --gpus 4 --transformation resize --resolution 8K --steps 32 --warmup 8 --segments 2 --batch 32 --no-compress > $O/${R}_bench_4ranks_on_1gpu_8K.json 2>> $O/ranks.err
  $B --gpus 8 --steps 20 --warmup 5 > $O/${R}_bench_8ranks_on_1gpu_driver_window.json 2>> $O/ranks.err      # (eight processes on the one GPU: the eight-rank tree end to end, nothing about speed)
  $B --gpus 8 --steps 64 --warmup 8 > $O/${R}_bench_8ranks_on_1gpu_w64.json 2>> $O/ranks.err
  $B --gpus 2 --mode accumulator > $O/${R}_bench_2ranks_on_1gpu_accumulator.json 2>> $O/ranks.err
  $B --gpus 4 --no-compress --proof-set contrast,brightness,sharpness,blur --resolution 4K --steps 48 --warmup 8 > $O/${R}_bench_proof_set_4K_4ranks_on_1gpu.json 2>> $O/ranks.err
  timeout 300 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/${R}_bench_plain_gpus2_on_1gpu.json 2> $O/${R}_bench_plain_gpus2_on_1gpu.txt; echo "exit code $?" >> $O/${R}_bench_plain_gpus2_on_1gpu.txt
fi
if want cores; then    # what a rank has when the ranks of a node share a small CPU quota: the same bench on 2 / 4 host cores (sched_setaffinity before anything starts)
  : > $O/${R}_cores.txt
  for c in 2 4 8; do for seg in 1 2 3; do
    timeout 900 python3 bench.py --cores $c --segments $seg --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cores $c segments $seg: %.1f steps/s over 256 rows, verified %s, host phases %s' % (d['value'], d['verified'], {k: round(v, 3) for k, v in d['phase_ms_per_step_per_proof'].items() if 'host' in k}))" >> $O/${R}_cores.txt
  done; done
  timeout 900 python3 bench.py --cores 2 --no-extras --no-cpu-baseline > $O/${R}_bench_2cores.json 2>/dev/null
  timeout 900 python3 bench.py --cores 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/${R}_bench_2cores_driver_window.json 2>/dev/null
fi
if want cyclefold; then  # the Sonobe backend's path: Nova + CycleFold on one chain
  : > $O/${R}_cyclefold.jsonl
  timeout 900 python3 tools/cyclefold_bench.py contrast HD 256 2>/dev/null | tail -1 >> $O/${R}_cyclefold.jsonl
  for cfg in "contrast HD 1" "contrast HD 3" "grayscale HD 3" "crop HD 3" "contrast 4K 1" "contrast 4K 3" "resize 8K 3"; do
    timeout 900 python3 tools/e2e.py $cfg cyclefold 2>/dev/null | tail -1 >> $O/${R}_cyclefold.jsonl
  done
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 tools/cyclefold_bench.py contrast HD 128 > $O/${R}_cyclefold_under_rocprof.json 2> $O/rocprof.err
  cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${R}_kernel_stats_cyclefold_HD.csv
  python3 tools/trace_busy.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/${R}_trace_busy_cyclefold_HD.txt 2>/dev/null
  rm -rf $O/kt
  python3 tools/small_msm_crossover.py > $O/${R}_small_msm_crossover.txt 2>/dev/null
fi
if want round5; then   # round 5's own evidence: the large MSM's tails under stress, the plane reduce A/B, set-up split, memory against batch size, the T-multiplicity experiment
  $T python3 tools/stress_large_msm.py 305185 30 3 > $O/${R}_stress_large_msm.txt 2>&1
  { echo "# shared bucket set reduced as virtual windows (k_reduce, default) vs by bit planes (VIMZ_TUNE=reduce_planes=1): alone on the GPU, then inside the bench";
    python3 tools/msm_bench.py 305185 2>&1 | grep tables; VIMZ_TUNE=reduce_planes=1 python3 tools/msm_bench.py 305185 2>&1 | grep tables | sed 's/^/planes: /';
    for pl in 0 1; do for args in "--steps 256 --repeats 3 --no-extras" "--steps 20 --warmup 5"; do
      VIMZ_TUNE=reduce_planes=$pl timeout 600 python3 bench.py --no-cpu-baseline --no-compress $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reduce_planes=$pl $args: %.0f steps/s, samples %s, one chain %s, reduce %.3f ms in the bench' % (d['value'], [round(x) for x in d['samples_steps_per_s']], round(d['one_chain']['steps_per_s']) if d.get('one_chain') else None, d['roofline']['msm_phase_ms']['reduce']))"
    done; done; } > $O/${R}_reduce_planes.txt
  : > $O/${R}_setup_breakdown.jsonl
  for cfg in "contrast HD 3" "contrast 4K 3" "resize 8K 3"; do timeout 600 python3 tools/setup_breakdown.py $cfg 2>/dev/null | tail -1 >> $O/${R}_setup_breakdown.jsonl; done
  { echo "# steps/s and peak device memory against the witness batch size (three provers per GPU; bench.py --batch B)";
    for cfg in "contrast HD 256" "contrast 4K 128" "resize 8K 128"; do set -- $cfg; for b in 32 64 128; do
      timeout 900 python3 bench.py --transformation $1 --resolution $2 --steps $3 --warmup 32 --repeats 3 --batch $b --no-extras --no-cpu-baseline --no-compress 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 batch $b: %.0f steps/s over $3 rows, peak device memory %.1f GB, verified %s' % (d['value'], d['peak_device_bytes'] / 1e9, d['verified']))"
    done; done; } > $O/${R}_batch_sweep.txt
  timeout 900 python3 tools/t_multiplicity.py 256 150 > $O/${R}_t_multiplicity.json 2> $O/t_mult.err
fi
if want round6; then   # round 6's own evidence: the full and the light decider end to end, the boolean-row form of the cross term (gate count, on / off on the same box)
  (python3 tools/e2e.py contrast HD 1 cyclefold 2>/dev/null | tail -1; VIMZ_E2E_DECIDER=light python3 tools/e2e.py contrast HD 1 cyclefold 2>/dev/null | tail -1;
   python3 tools/e2e.py hash HD 1 cyclefold 2>/dev/null | tail -1; python3 tools/e2e.py blur HD 1 cyclefold 2>/dev/null | tail -1) > $O/${R}_e2e_decider.jsonl
  timeout 900 python3 tools/t_boolean_rows.py 256 150 > $O/${R}_t_boolean_rows.json 2> $O/t_bool.err
  { echo "# the boolean-row form of the step rows' cross-term commitment (VIMZ_IVC_BOOL_ROWS=1, default) against the plain vector (=0), same box, same commands";
    for v in 1 0; do for args in "--steps 20 --warmup 5" "--steps 256 --warmup 16" "--steps 720 --warmup 16 --repeats 3"; do
      VIMZ_IVC_BOOL_ROWS=$v timeout 600 python3 bench.py --no-cpu-baseline --no-extras --no-compress $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bool_rows=$v $args: %.0f steps/s, samples %s, MSM(T) phases in the bench %s, bucket additions per launch %.0f' % (d['value'], [round(x) for x in d['samples_steps_per_s']], {k: round(x, 3) for k, x in d['roofline']['msm_phase_ms'].items()}, d['roofline']['mixed_adds_per_launch']))"
    done; done
    for v in 1 0; do for cfg in "contrast HD 3 ivc" "crop HD 3 ivc" "contrast 4K 3 ivc"; do
      VIMZ_IVC_BOOL_ROWS=$v timeout 600 python3 tools/e2e.py $cfg 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bool_rows=$v whole image', d['config'], '%.0f steps/s, fold %.3f s, prepare folding %.3f s, wall %.2f s' % (d['steps_per_s'], d['spans_s']['Fold input'], d['spans_s']['Prepare folding'], d['wall_total_s']))"
    done; done; } > $O/${R}_bool_rows_ab.txt
fi
if want stalls; then   # k_accum alone on the GPU (tools/msm_bench.py: 305 185 dense scalars): where its wave cycles go, and the bytes it fetches per launch (64-byte table entries)
  $T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/ps -o ps -- python3 tools/msm_bench.py 305185 > /dev/null 2>> $O/rocprof.err
  { echo "# k_accum over window tables, alone on the GPU (msm_bench.py 305185): SQ counters per launch, by grid size"; python3 tools/stall_summary.py $(find $O/ps -name "*counter_collection.csv" | head -1) k_accum; } > $O/${R}_k_accum_stalls_alone.txt
  rm -rf $O/ps
  $T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o pf -- python3 tools/msm_bench.py 305185 > /dev/null 2>> $O/rocprof.err
  { echo "# FETCH_SIZE per k_accum launch, alone (units as rocprofv3 reports them; pmc_summary.py applies the guide's correction for the bench passes)"; python3 tools/stall_summary.py $(find $O/pf -name "*counter_collection.csv" | head -1) k_accum; } >> $O/${R}_k_accum_stalls_alone.txt
  rm -rf $O/pf
fi
ls -la $O
