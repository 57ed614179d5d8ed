#!/bin/bash
# Launch policy of the march per launch kind, with counters: march time and HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, each
# counter in its own rocprofv3 pass) for {persistent waves with 1 / 4 sub-queues per XCD} x {whole marches, 8 segments} on C3
# (coherent BOS cones), C5 at a quarter (incoherent, lens-major) and C4 whole (512^3: 4 GiB of texels).  (Round 4's table,
# profiles/r04_policy_table.json, also has a one-shot grid column: that code path was removed in round 5.)
#   tools/policy_table.sh [workload ...]        variants: build/variants/lib_sq1.so (-DPHOTON_SUBQUEUES=1) and the in-tree library
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
export TMPDIR=/tmp
wl=("$@"); [ ${#wl[@]} -eq 0 ] && wl=(c3 c5 c4)
out=$ROOT/gpurun_out/policy; mkdir -p "$out"
B="$ROOT/bench.py --steps 3 --warmup 1 --cpu-sample-rays 0 --no-traffic --no-other-configs --no-profile"
for w in "${wl[@]}"; do
  case $w in
    c3) cmd="$B" ;;
    c4) cmd="$B --volume 512 --dots 2000" ;;
    c5) cmd="$ROOT/tools/c5_full.py 0.25" ;;
  esac
  for v in sq1 default; do
    for seg in 1 8; do
      if [ $v = default ]; then unset PHOTON_LIBRARY; else export PHOTON_LIBRARY=$ROOT/build/variants/lib_$v.so; fi
      export PHOTON_MARCH_SEGMENTS=$seg
      tag=${w}_${v}_s$seg
      python3 $cmd > "$out/$tag.time.log" 2>&1 || { echo "$tag FAILED"; tail -3 "$out/$tag.time.log"; continue; }
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && rocprofv3 --pmc $c --output-format csv -d "$out/$tag.$c" -o p -- python3 $cmd > "$out/$tag.$c.log" 2>&1)
      done
      python3 - "$out" "$tag" <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
line = json.loads([l for l in open(f"{out}/{tag}.time.log") if l.startswith("{")][-1])
ms = line["roofline"]["kernel_ms"] if "roofline" in line else line["march_ms"]
clock = (line.get("roofline") or {}).get("clock_mhz")
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    xs = [float(r["Counter_Value"]) for f in glob.glob(f"{out}/{tag}.{c}/**/*_counter_collection.csv", recursive=True)
          for r in csv.DictReader(open(f)) if "march_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    vals[c] = sum(xs) / len(xs) if xs else float("nan")
gb = (vals["FETCH_SIZE"] * 2 + vals["WRITE_SIZE"]) * 1024 / 1e9
print(json.dumps({"case": tag, "march_ms": round(ms, 3), "clock_mhz": clock, "hbm_GB_per_launch": round(gb, 3),
                  "FETCH_KiB": round(vals["FETCH_SIZE"]), "WRITE_KiB": round(vals["WRITE_SIZE"])}), flush=True)
PY
      rm -rf "$out/$tag.FETCH_SIZE" "$out/$tag.WRITE_SIZE"
    done
  done
done
