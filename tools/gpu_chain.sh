#!/bin/bash
# Run GPU steps one after another on a gpurun box: each step under its own `timeout -k 10`, logs under gpurun_out/.
# A step that FAILS (test assertion, exit code below 128) does not stop the chain; a step that TIMES OUT, is KILLED or dies of a
# signal (rc >= 124: 124 timeout, 134 SIGABRT, 137 SIGKILL, 139 SIGSEGV - how a GPU fault or HIP abort surfaces) does: after a
# hang or a fault no further GPU step may start in the same call.
#   tools/gpu_chain.sh "<secs> <logname> <command...>" ...
mkdir -p gpurun_out
for spec in "$@"; do
  secs=${spec%% *}; rest=${spec#* }; log=${rest%% *}; cmd=${rest#* }
  echo "[chain] $log: $cmd (limit ${secs}s)"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$log" 2>&1
  rc=$?
  echo "[chain] $log rc=$rc"; tail -n 6 "gpurun_out/$log"
  if [ $rc -ge 124 ]; then echo "[chain] step timed out / was killed / died of a signal: stopping"; tail -n 30 "gpurun_out/$log"; exit $rc; fi
done
exit 0
