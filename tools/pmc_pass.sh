#!/bin/bash
# One rocprofv3 --pmc pass over the default bench command; prints per-launch means for the march kernels.
#   tools/pmc_pass.sh <tag> COUNTER [COUNTER...]        (optionally PHOTON_LIBRARY=... in the environment)
set -e
tag=$1; shift
out=$PWD/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d "$out" -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample-rays 0 $PHOTON_BENCH_ARGS > "$out/bench.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 - "$out" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "march_kernel" in k:
        print(k, {c: f"{sum(x) / len(x):.4g}" for c, x in v.items()})
PY
