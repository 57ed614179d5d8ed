#!/bin/bash
# PMC counters of the march kernel for one library variant on a bench workload:
#   tools/pmc_variant.sh <variant|default> "<bench args>" COUNTER [COUNTER ...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
n=$1; args=$2; shift 2
if [ "$n" = default ]; then unset PHOTON_LIBRARY; else export PHOTON_LIBRARY=$ROOT/build/variants/lib_$n.so; fi
export TMPDIR=/tmp
out=$ROOT/gpurun_out/pmcv_$n; rm -rf "$out"; mkdir -p "$out"
(cd /tmp && rocprofv3 --pmc "$@" --output-format csv -d "$out" -o p -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-sample-rays 0 --no-traffic $args > "$out/run.log" 2>&1)
python3 - "$out" "$n" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "march_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {c: f"{sum(x) / len(x):.4g}" for c, x in acc.items()})
PY
