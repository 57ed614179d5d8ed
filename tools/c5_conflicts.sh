#!/bin/bash
# SQ_LDS_BANK_CONFLICT of the incoherent (brick) launch, per library variant: tools/c5_conflicts.sh <variant|default> ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" = default ]; then unset PHOTON_LIBRARY; else export PHOTON_LIBRARY=$ROOT/build/variants/lib_$n.so; fi
  out=$ROOT/gpurun_out/c5pmc_$n; rm -rf "$out"; mkdir -p "$out"
  (cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$out" -o p -- python3 $ROOT/tools/c5_full.py 0.25 > "$out/run.log" 2>&1)
  python3 - "$out" "$n" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "march_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {c: f"{sum(x) / len(x):.4g}" for c, x in acc.items()})
PY
done
