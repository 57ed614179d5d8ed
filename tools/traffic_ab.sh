#!/bin/bash
# HBM traffic (bench.py's own rocprofv3 --pmc passes), march time, clock and power per library variant:
#   tools/traffic_ab.sh "<bench args>" <variant|default> ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
args=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset PHOTON_LIBRARY; else export PHOTON_LIBRARY=$ROOT/build/variants/lib_$v.so; fi
  timeout -k 10 300 python bench.py --steps 5 --cpu-sample-rays 0 $args > gpurun_out/traffic_$v.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/traffic_$v.log; continue; }
  python - gpurun_out/traffic_$v.log "$v" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print(sys.argv[2], "march_ms", r["kernel_ms"], "traffic_GB", round((r["traffic"] or 0) / 1e9, 2), "clock", r.get("clock_mhz"),
      "frac_at_clock", r.get("frac_at_clock"), "power_w", (r.get("board_power") or {}).get("median_w"))
PY
done
