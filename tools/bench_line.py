"""One line per bench log: the numbers A/B runs are compared on.   python tools/bench_line.py <log> [...]"""
import json
import sys

for path in sys.argv[1:]:
    for ln in open(path):
        if not ln.startswith("{"):
            continue
        d = json.loads(ln)
        r = d["roofline"]
        t = r.get("algorithmic_texel_rate") or r.get("texel_rate_vs_lds") or r      # (round-4/5 name; before round 4 at the top level)
        p = r.get("march_profile") or {}
        print(path, "Mrays/s", d["value"], "ms/step", d["ms_per_step"], "march_ms", r["kernel_ms"], "clock_mhz", r.get("clock_mhz"),
              "issue_frac", r.get("frac"), "texel_GBs", t.get("achieved"), "wave_ms", r.get("wave_lifetime_ms"),
              "gens", r.get("wave_generations"), "span/drain_ms", p.get("span_ms"), p.get("drain_ms"),
              "power_w", (r.get("board_power") or {}).get("median_w"), "check", d.get("check"))
