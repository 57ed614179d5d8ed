#!/usr/bin/env python3
"""Print the key numbers of bench.py JSON lines: tools/bench_line.py LOG [LOG...]"""
import json
import sys

for path in sys.argv[1:]:
    try:
        lines = [x for x in open(path) if x.startswith("{")]
        d = json.loads(lines[-1])
        r = d["roofline"]
        print(path, "Mrays/s", d["value"], "ms/step", d["ms_per_step"], "march_ms", r["kernel_ms"], "frac", r.get("frac"),
              "clock_mhz", r.get("clock_mhz"), "frac_at_clock", r.get("frac_at_clock"), "wave_ms", r.get("wave_lifetime_ms"), "gens", r.get("wave_generations"), "power_w", (r.get("board_power") or {}).get("median_w"), "cap", (r.get("board_power") or {}).get("cap_w"), "check", d.get("check"))
    except Exception as e:                                     # noqa: BLE001
        print(path, "unreadable:", e)
