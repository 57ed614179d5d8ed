#!/bin/bash
# March time against the number of segments (forced), full C3 job and one GPU's eighth:  tools/segments_sweep.sh [interp]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
interp=${1:-cubic}
mkdir -p gpurun_out/sweep
for dots in 200 25; do
  steps=20; [ $dots = 25 ] && steps=80
  for s in 1 2 3 4 6 8 12 16 24; do
    PHOTON_MARCH_SEGMENTS=force:$s timeout -k 10 120 python bench.py --interp $interp --dots $dots --steps $steps --warmup 3 --cpu-sample-rays 0 --no-traffic --no-other-configs > gpurun_out/sweep/${interp}_${dots}_$s.log 2>&1 || { echo "$dots $s FAILED"; continue; }
    echo -n "$interp dots $dots segments $s: "; python tools/bench_line.py gpurun_out/sweep/${interp}_${dots}_$s.log | cut -d' ' -f2-22
  done
done
