#!/usr/bin/env python3
"""Condense tools/profile_round.sh output into profiles/: per workload the kernel stats CSV (photon kernels only) and one
JSON with the per-launch PMC means of the march / sensor / splat kernels.   profile_round_summary.py <dir> <tag>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ("march", "sensor", "splat", "raygen", "finalize", "morton", "bbox", "RadixSort", "radix", "postprocess")


def main():
    src, tag = sys.argv[1], sys.argv[2]
    dst = os.path.join(ROOT, "profiles")
    for wdir in sorted(glob.glob(os.path.join(src, "*"))):
        if not os.path.isdir(wdir):
            continue
        w = os.path.basename(wdir)
        rows = []
        for f in glob.glob(os.path.join(wdir, "stats", "**", "*_kernel_stats.csv"), recursive=True):
            with open(f) as fh:
                rd = csv.DictReader(fh)
                fields = rd.fieldnames
                rows += [r for r in rd if any(k in r["Name"] for k in KEEP)]
        if rows:
            with open(os.path.join(dst, f"{tag}_{w}_kernel_stats.csv"), "w", newline="") as fh:
                wr = csv.DictWriter(fh, fieldnames=fields)
                wr.writeheader()
                wr.writerows(sorted(rows, key=lambda r: -float(r["TotalDurationNs"])))
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(wdir, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    name = r["Kernel_Name"].split("(")[0]
                    if any(k in name for k in KEEP[:4]):
                        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        pmc = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches_sampled": max(len(v) for v in cs.values())} for k, cs in acc.items()}
        for k, cs in pmc.items():
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                cs["hbm_bytes_per_launch"] = int(cs["FETCH_SIZE"] * 1024 * 2 + cs["WRITE_SIZE"] * 1024)    # guide: FETCH x2 on gfx950
        line = None
        for log in glob.glob(os.path.join(wdir, "stats.log")):
            with open(log) as fh:
                for ln in fh:
                    if ln.startswith("{"):
                        line = json.loads(ln)
        with open(os.path.join(dst, f"{tag}_{w}_pmc.json"), "w") as fh:
            json.dump({"workload": w, "bench_line_under_kernel_trace": line, "per_launch_means": pmc}, fh, indent=1)
        for r in rows[:0]:
            pass
        print(w, {r["Name"].split("(")[0][:40]: round(float(r["AverageNs"]) / 1e6, 3) for r in rows if "march" in r["Name"] or "splat" in r["Name"]})


if __name__ == "__main__":
    main()
