#!/bin/bash
# Regenerates every measurement committed under profiles/ on the GPU box (one gpurun call):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r01'
# then, back in the container:  python tools/summarize_profiles.py ...  (printed at the end)
set -u
R=${1:-r02}
export TMPDIR=/tmp
O=gpurun_out/$R
mkdir -p $O
# 1. the bench line (default: f16x3 engine, one batch in flight = `value`; + multi_stream, bf16_engine, cpu_baseline side lines)
timeout -k 10 600 python bench.py > $O/bench_r50.json 2> $O/bench_r50.err
# 2. kernel trace + stats of ONE handle (what roofline.avg_launch_us must agree with), then the PMC passes.  side_stream=0: every kernel of the
#    step on one stream, so that durations and counters belong to one kernel at a time (the product overlaps the query-selection chain with the
#    value projection on a second stream: -40 us per step, but overlapped kernels' durations would be counted twice)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o runc -- python3 bench.py --steps 20 --warmup 3 --streams 1 --multi-streams 0 --no-bf16-line --no-cpu-baseline --no-latency --opt side_stream=0 > $O/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o runc -- python3 bench.py --steps 3 --warmup 1 --streams 1 --multi-streams 0 --no-bf16-line --no-cpu-baseline --no-latency --opt side_stream=0 > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o runc -- python3 bench.py --steps 3 --warmup 1 --streams 1 --multi-streams 0 --no-bf16-line --no-cpu-baseline --no-latency --opt side_stream=0 > $O/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o runc -- python3 bench.py --steps 3 --warmup 1 --streams 1 --multi-streams 0 --no-bf16-line --no-cpu-baseline --no-latency --opt side_stream=0 > $O/pmc_mfma.log 2>&1
# 3. the other configurations: R18 (the reference's default model), the exact fp32 engine, BASELINE config 3 (R101 1280^2 bs 4)
timeout -k 10 300 python bench.py --arch r18 --no-cpu-baseline > $O/bench_r18.json 2>/dev/null
timeout -k 10 300 python bench.py --precision bf16 --no-cpu-baseline > $O/bench_bf16.json 2>/dev/null
timeout -k 10 300 python bench.py --precision fp32 --no-cpu-baseline --steps 30 --warmup 5 --multi-streams 0 > $O/bench_fp32.json 2>/dev/null
timeout -k 10 400 python bench.py --arch r101 --size 1280 --batch 4 --steps 30 --warmup 5 --multi-streams 2 --no-cpu-baseline > $O/bench_r101.json 2>/dev/null
timeout -k 10 300 python bench.py --workload two_stage --no-cpu-baseline --no-latency > $O/bench_two_stage.json 2>/dev/null
# 4. per-layer HIP-event profile of one eager forward
timeout -k 10 200 python tools/profile_layers.py --out $O/layers_r50.json > $O/layers_r50.log 2>&1
find $O -name "*.csv" | head -20
