#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench for one library: tools/kernel_times.sh <tag> [bench args]
set -e
tag=$1; shift
out=$PWD/gpurun_out/kt_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample-rays 0 "$@" > "$out/bench.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("march", "sensor", "raygen")):
            print("  ", r["Name"][:44], r["Calls"], round(float(r["AverageNs"]) / 1e6, 3), "ms")
PY
