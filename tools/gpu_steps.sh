#!/bin/bash
# Run a list of GPU steps in ONE gpurun call:   tools/gpu_steps.sh <outdir> "<name>|<seconds>|<command>" ...
# Each step runs under `timeout -k 10`, its output goes to gpurun_out/<outdir>/<name>.log (a progress line is echoed, so the
# call never looks silent); a step that fails goes on to the next one, a step that TIMES OUT or is killed ends the call --
# nothing further is started on a GPU that may be hung.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
out=gpurun_out/$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
for step in "$@"; do
  name=${step%%|*}; rest=${step#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "== $(date +%T) $name: $cmd"
  ( timeout -k 10 "$secs" bash -c "$cmd" ) > "$out/$name.log" 2>&1 &
  pid=$!
  while kill -0 $pid 2>/dev/null; do sleep 20; echo "   ... $name running ($(wc -c < "$out/$name.log") bytes of output)"; done
  wait $pid; rc=$?
  echo "== $name rc=$rc"; tail -4 "$out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name timed out: stopping here"; exit $rc; fi
done
exit 0
