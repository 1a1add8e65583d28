#!/bin/bash
# Same-box rocprofv3 kernel traces of two builds of the library (tools/_bin/lib_old.so = an earlier build, see tools/ab_lib.sh): the
# profiled step of each, one after the other in ONE gpurun call, so that the conv family's microseconds per step compare.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/ab_rocprof.sh [extra bench.py flags]'
export TMPDIR=/tmp
O=gpurun_out/ab_rocprof
mkdir -p $O
PROF="--steps 20 --warmup 3 --streams 1 --multi-streams 0 --no-bf16-line --no-cpu-baseline --no-latency --no-detect-host --no-mfma-probe --opt side_stream=0"
for rep in 1 2; do
  for which in old new; do
    if [ $which = old ]; then export RTD_LIB_PATH=$PWD/tools/_bin/lib_old.so; else unset RTD_LIB_PATH; fi
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${which}_$rep -o runc -- python3 bench.py $PROF "$@" > $O/${which}_$rep.log 2> $O/${which}_$rep.err
    rc=$?
    echo "[ab_rocprof] $which $rep rc=$rc"
    if [ $rc -ge 124 ]; then exit $rc; fi
  done
done
