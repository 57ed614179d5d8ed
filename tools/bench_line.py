#!/usr/bin/env python3
"""Print the key numbers of bench.py JSON lines: tools/bench_line.py LOG [LOG...]"""
import json
import sys

for path in sys.argv[1:]:
    try:
        lines = [x for x in open(path) if x.startswith("{")]
        d = json.loads(lines[-1])
        r = d["roofline"]
        print(path, "Mrays/s", d["value"], "ms/step", d["ms_per_step"], "march_ms", r["kernel_ms"],
              "valu", r.get("valu_f32", {}).get("frac"), "copy GB/s", r.get("hbm_copy_measured_gbs"))
    except Exception as e:                                     # noqa: BLE001
        print(path, "unreadable:", e)
