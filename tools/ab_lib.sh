#!/bin/bash
# Same-box A/B of two builds of the library (boxes of the pool differ by several per cent in the clock they hold under MFMA load, so
# two gpurun calls never compare): tools/_bin/lib_old.so (a copy of an earlier telescope_cam_detection_amd/lib/libmi355rtdetr.so) against
# the current build, alternating, bench.py's `value` and the conv family's per-step time.
#   cp telescope_cam_detection_amd/lib/libmi355rtdetr.so tools/_bin/lib_old.so   (before rebuilding)
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/ab_lib.sh [extra bench.py flags]'
B="--steps 100 --warmup 10 --multi-streams 0 --no-bf16-line --no-cpu-baseline --no-latency --no-detect-host"
for rep in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then export RTD_LIB_PATH=$PWD/tools/_bin/lib_old.so; else unset RTD_LIB_PATH; fi
    timeout -k 10 200 python bench.py $B "$@" 2>/dev/null | tail -1 > /tmp/ab_$which.json || exit 1
    python - $which <<'PY'
import json, sys
d = json.load(open(f"/tmp/ab_{sys.argv[1]}.json"))
print(f"{sys.argv[1]:4s}: {d['value']:8.1f} frames/s  {d['ms_per_step']:.4f} ms/step  conv {d['kernel_families_ms']['conv_igemm']:.4f} ms  dec {d['kernel_families_ms'].get('dec_layer', 0):.4f} ms", flush=True)
PY
  done
done
