#!/bin/bash
# Turns what tools/refresh_profiles.sh left under gpurun_out/<round>/ into the committed summaries under profiles/ (run in the build
# container, kernel sources unchanged since the GPU runs: the summaries are stamped with the csrc hash bench.py checks).
#   bash tools/commit_profiles.sh r03
set -eu
R=${1:-r05}
T=gpurun_out/$R; P=profiles/$R
f() { ls $1 2>/dev/null | head -1; }
if [ -n "$(f "$T/trace/*kernel_trace.csv")" ]; then
  python tools/summarize_profiles.py trace $(f "$T/trace/*kernel_trace.csv") ${P}_rocprofv3_kernel_summary.json
  cp $(f "$T/trace/*kernel_stats.csv") ${P}_rocprofv3_kernel_stats.csv
  python tools/summarize_profiles.py pmc $(f "$T/pmc_fetch/*counter_collection.csv") $(f "$T/pmc_write/*counter_collection.csv") ${P}_pmc_hbm_traffic.json
  python tools/summarize_profiles.py mfma $(f "$T/pmc_mfma/*counter_collection.csv") ${P}_pmc_mfma_util.json
  cp $T/layers_r50.json ${P}_layers_r50_bs8_f16x3.json
  tail -n 1 $T/bench_r50.json > ${P}_bench_r50_bs8_f16x3.json
fi
if [ -n "$(f "$T/trace_r101/*kernel_trace.csv")" ]; then
  export RTD_PROFILE_CONFIG=r101_1280_bs4_f16x3 RTD_PROFILE_ARGS="--arch r101 --size 1280 --batch 4"
  python tools/summarize_profiles.py trace $(f "$T/trace_r101/*kernel_trace.csv") ${P}_rocprofv3_kernel_summary_r101_1280_bs4.json
  cp $(f "$T/trace_r101/*kernel_stats.csv") ${P}_rocprofv3_kernel_stats_r101_1280_bs4.csv
  python tools/summarize_profiles.py pmc $(f "$T/pmc_fetch_r101/*counter_collection.csv") $(f "$T/pmc_write_r101/*counter_collection.csv") ${P}_pmc_hbm_traffic_r101_1280_bs4.json
  python tools/summarize_profiles.py mfma $(f "$T/pmc_mfma_r101/*counter_collection.csv") ${P}_pmc_mfma_util_r101_1280_bs4.json
  cp $T/layers_r101.json ${P}_layers_r101_1280_bs4_f16x3.json
  tail -n 1 $T/bench_r101.json > ${P}_bench_r101_1280_bs4_f16x3.json
  unset RTD_PROFILE_CONFIG RTD_PROFILE_ARGS
fi
for n in r18:r18_bs8_f16x3 bf16:r50_bs8_bf16 fp32:r50_bs8_fp32 two_stage:two_stage_r50_bs8_f16x3 collate:collate_r50_bs8_f16x3 c4:c4_r50_bs1_collate; do
  src=$T/bench_${n%%:*}.json
  if [ -s $src ]; then tail -n 1 $src > ${P}_bench_${n##*:}.json; fi
done
ls -la profiles | grep $R
