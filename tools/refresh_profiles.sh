#!/bin/bash
# Regenerates every measurement committed under profiles/ on the GPU box (one gpurun call each; a part that times out stops the script):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r05 r50'      (default config: R50 640^2 bs 8, f16x3)
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r05 r101'     (BASELINE configs[2]: R101 1280^2 bs 4)
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r05 lines'    (the other bench lines)
# then, back in the container, tools/summarize_profiles.py turns the CSVs into profiles/<round>_*.json (commands at the end of this file)
set -u
R=${1:-r05}
WHAT=${2:-r50}
export TMPDIR=/tmp
O=gpurun_out/$R
mkdir -p $O
run() {   # run <seconds> <log> <command...>: stop the whole script when a step times out, is killed or dies of a signal
  local secs=$1 log=$2; shift 2
  timeout -k 10 "$secs" "$@" > "$log" 2> "${log%.*}.err"
  local rc=$?
  echo "[refresh] $log rc=$rc"
  if [ $rc -ge 124 ]; then echo "[refresh] step timed out / died: stopping"; tail -n 20 "${log%.*}.err"; exit $rc; fi
}
PROF="--streams 1 --multi-streams 0 --no-bf16-line --no-cpu-baseline --no-latency --no-detect-host --no-mfma-probe --opt side_stream=0"
if [ "$WHAT" = r50 ]; then
  # 1. the bench line (f16x3 engine, one batch in flight = `value`; + multi_stream, bf16_engine, detect_host_ms, cpu_baseline side lines)
  run 900 $O/bench_r50.json python bench.py
  # 2. kernel trace + stats of ONE handle (what roofline.avg_launch_us must agree with), then the PMC passes.  side_stream=0: every kernel of the
  #    step on one stream, so that durations and counters belong to one kernel at a time (the product overlaps the query-selection chain with the
  #    value projection on a second stream, but overlapped kernels' durations would be counted twice)
  run 300 $O/trace.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o runc -- python3 bench.py --steps 20 --warmup 3 $PROF
  run 300 $O/pmc_fetch.log rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o runc -- python3 bench.py --steps 3 --warmup 1 $PROF
  run 300 $O/pmc_write.log rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o runc -- python3 bench.py --steps 3 --warmup 1 $PROF
  run 300 $O/pmc_mfma.log rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o runc -- python3 bench.py --steps 3 --warmup 1 $PROF
  run 200 $O/layers_r50.log python tools/profile_layers.py --out $O/layers_r50.json
elif [ "$WHAT" = r101 ]; then
  A="--arch r101 --size 1280 --batch 4"
  run 600 $O/bench_r101.json python bench.py $A --steps 30 --warmup 5 --multi-streams 2 --no-cpu-baseline --no-detect-host
  run 300 $O/trace_r101.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_r101 -o runc -- python3 bench.py $A --steps 10 --warmup 3 $PROF
  run 300 $O/pmc_fetch_r101.log rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_r101 -o runc -- python3 bench.py $A --steps 3 --warmup 1 $PROF
  run 300 $O/pmc_write_r101.log rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_r101 -o runc -- python3 bench.py $A --steps 3 --warmup 1 $PROF
  run 300 $O/pmc_mfma_r101.log rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_r101 -o runc -- python3 bench.py $A --steps 3 --warmup 1 $PROF
  run 300 $O/layers_r101.log python tools/profile_layers.py --arch r101 --size 1280 --batch 4 --out $O/layers_r101.json
else
  # the other configurations: R18 (the reference's default model), plain bf16 and exact fp32 engines, Stage 2, RCCL collate at world size 1
  run 300 $O/bench_r18.json python bench.py --arch r18 --no-cpu-baseline --no-detect-host
  run 300 $O/bench_bf16.json python bench.py --precision bf16 --no-cpu-baseline
  run 300 $O/bench_fp32.json python bench.py --precision fp32 --no-cpu-baseline --steps 30 --warmup 5 --multi-streams 0
  run 300 $O/bench_two_stage.json python bench.py --workload two_stage --no-cpu-baseline --no-latency
  run 300 $O/bench_collate.json python bench.py --collate --no-cpu-baseline --no-latency --no-detect-host --multi-streams 0 --no-bf16-line
  # BASELINE configs[3]'s per-rank shape (one camera per GPU: bs 1 + the collate all-gather) at N = 1: the line a SCALE run's N = 1 must agree with
  run 300 $O/bench_c4.json python bench.py --batch 1 --collate --no-cpu-baseline --no-detect-host --multi-streams 0 --no-bf16-line
fi
find $O -name "*.csv" | head -40
# back in the container (kernel sources unchanged since the run):
#   T=gpurun_out/r04; P=profiles/r04
#   python tools/summarize_profiles.py trace $(ls $T/trace/runc/*kernel_trace.csv) ${P}_rocprofv3_kernel_summary.json
#   python tools/summarize_profiles.py pmc $(ls $T/pmc_fetch/runc/*counter_collection.csv) $(ls $T/pmc_write/runc/*counter_collection.csv) ${P}_pmc_hbm_traffic.json
#   python tools/summarize_profiles.py mfma $(ls $T/pmc_mfma/runc/*counter_collection.csv) ${P}_pmc_mfma_util.json
#   RTD_PROFILE_CONFIG=r101_1280_bs4_f16x3 RTD_PROFILE_ARGS="--arch r101 --size 1280 --batch 4" python tools/summarize_profiles.py trace ... ${P}_rocprofv3_kernel_summary_r101_1280_bs4.json   (etc.)
