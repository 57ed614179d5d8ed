#!/bin/bash
# Same-box A/B of ONE library under two settings of an environment variable, alternating runs:
#   tools/ab_env.sh <VAR> <value A> <value B> <rounds> "<bench args>" [...more bench arg sets]
# One line per run (tools/bench_line.py); logs under gpurun_out/abenv_*.log.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
var=$1; a=$2; b=$3; rounds=$4; shift 4
mkdir -p gpurun_out
for args in "$@"; do
  tag=$(echo "$args" | tr -c 'A-Za-z0-9' '_')
  for ((i = 1; i <= rounds; i++)); do
    for v in "$a" "$b"; do
      log=gpurun_out/abenv_${var}_${v}_${tag}_$i.log
      env "$var=$v" timeout -k 10 180 python bench.py --cpu-sample-rays 0 --no-traffic --no-other-configs --no-profile $args > "$log" 2>&1 || { echo "$v FAILED"; tail -5 "$log"; exit 1; }
      echo -n "[$args] $var=$v: "; python tools/bench_line.py "$log"
    done
  done
done
