#!/bin/bash
# March time against the number of segments (forced), full C3 job and one GPU's eighth:  tools/segments_sweep.sh [interp] [shape]
#   shape: uniform (equal pieces) | halving (1/2, 1/4, ... of the depth) | taper:<t> (equal pieces, the last one halved t times)
#   DOTS="200 25" picks the job sizes (200 = the full job, 25 = one GPU's eighth)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
interp=${1:-cubic}; shape=${2:-halving}
export PHOTON_MARCH_SEGMENT_SHAPE=$shape
mkdir -p gpurun_out/sweep
for dots in ${DOTS:-200 25}; do
  steps=20; [ $dots = 25 ] && steps=80
  list="1 2 3 4 5 6 7 8"; [ $shape != halving ] && list="1 2 3 4 6 8 12 16 24"
  [ -n "$LIST" ] && list=$LIST
  for s in auto $list; do
    if [ $s = auto ]; then unset PHOTON_MARCH_SEGMENTS; else export PHOTON_MARCH_SEGMENTS=force:$s; fi
    timeout -k 10 120 python bench.py --interp $interp --dots $dots --steps $steps --warmup 3 --cpu-sample-rays 0 --no-traffic --no-other-configs > gpurun_out/sweep/${shape/:/}_${interp}_${dots}_$s.log 2>&1 || { echo "$dots $s FAILED"; continue; }
    echo -n "$shape $interp dots $dots segments $s: "; python tools/bench_line.py gpurun_out/sweep/${shape/:/}_${interp}_${dots}_$s.log | cut -d' ' -f2-22
  done
done
