#!/bin/bash
# same-box A/B of one rtd_debug_option through bench.py:  tools/ab_bench.sh ws2_min_blocks 512 257
opt=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    python bench.py --steps 120 --warmup 12 --opt $opt=$v --no-cpu-baseline --no-latency 2>/dev/null > /tmp/ab.json
    python - "$opt" "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(f"{sys.argv[1]}={sys.argv[2]:>5s}: {d['value']:8.1f} frames/s ({d['config']['streams_per_gpu']} handles)  single {d['single_stream']['value']:8.1f}")
PY
  done
done
