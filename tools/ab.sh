#!/bin/bash
# Same-box A/B of library builds on the bench workloads: tools/ab.sh "<bench args>" <variant|default> ...   (order kept;
# name a variant twice to see the box's own repeatability).  A variant is build/variants/lib_<name>.so; `default` is
# the in-tree library.  One line per run (tools/bench_line.py); logs under gpurun_out/ab_*.log.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
args=$1; shift
mkdir -p gpurun_out
i=0
for n in "$@"; do
  i=$((i + 1))
  log=gpurun_out/ab_${i}_${n}.log
  if [ "$n" = default ]; then unset PHOTON_LIBRARY; else export PHOTON_LIBRARY=$ROOT/build/variants/lib_$n.so; fi
  timeout -k 10 180 python bench.py --steps 5 --warmup 2 --cpu-sample-rays 0 --no-traffic $args > "$log" 2>&1 || { echo "$n FAILED"; tail -5 "$log"; exit 1; }
  echo -n "[$args] $n: "; python tools/bench_line.py "$log"
done
