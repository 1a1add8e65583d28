#!/bin/bash
# same-box A/B of one rtd_debug_option through a lean bench.py (3 alternating repetitions):  tools/ab_opt.sh dead_out 0 1 [-- extra bench flags]
opt=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" = "--" ] && shift
B="--steps 200 --warmup 20 --multi-streams 0 --no-bf16-line --no-cpu-baseline --no-latency --no-detect-host --no-mfma-probe"
for rep in 1 2 3; do
  for v in "${vals[@]}"; do
    timeout -k 10 300 python bench.py $B --opt $opt=$v "$@" 2>/dev/null | tail -1 > /tmp/ab.json || exit 1
    python - "$opt" "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
f = d.get("kernel_families_ms", {})
print(f"{sys.argv[1]}={sys.argv[2]:>4s}: {d['value']:8.1f} frames/s  {d['ms_per_step']:.4f} ms/step  conv {f.get('conv_igemm', 0):.4f} ms  dec {f.get('dec_layer', 0):.4f} ms", flush=True)
PY
  done
done
