#!/usr/bin/env python3
"""Turn rocprofv3 --pmc CSV output into profiles/pmc_summary.json (what bench.py reports as
roofline.traffic).  HBM bytes per launch of the march kernel, corrected as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes for gfx950: FETCH_SIZE (KiB units)
x 2 for wide coalesced reads, WRITE_SIZE (KiB) as is; separate passes for the two counters.

    python tools/pmc_summary.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <key> [out.json]
"""
import collections
import csv
import glob
import json
import os
import sys


def kernel_means(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in out.items()}


def main():
    fetch_dir, write_dir, key = sys.argv[1:4]
    out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_summary.json")
    fm, wm = kernel_means(fetch_dir), kernel_means(write_dir)
    name = next(k for k in fm if "march_kernel" in k)
    fetch_kib, write_kib = fm[name]["FETCH_SIZE"], wm[name]["WRITE_SIZE"]
    entry = {"kernel": name, "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
             "hbm_bytes_per_launch": int(fetch_kib * 1024 * 2 + write_kib * 1024),
             "note": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads); WRITE_SIZE as counted"}
    data = {}
    if os.path.exists(out_path):
        with open(out_path) as f:
            data = json.load(f)
    data[key] = entry
    with open(out_path, "w") as f:
        json.dump(data, f, indent=1)
    print(json.dumps(entry))


if __name__ == "__main__":
    main()
