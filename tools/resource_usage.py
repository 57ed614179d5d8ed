#!/usr/bin/env python3
"""Tabulate hipcc's -Rpass-analysis=kernel-resource-usage remarks: one line per kernel (demangled name, VGPRs, SGPRs,
spills, scratch, occupancy, LDS), sorted by name -- so that two builds can be diffed.

    hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip -o x.o 2> remarks.txt
    python tools/resource_usage.py remarks.txt [more.txt ...]
"""
import re
import subprocess
import sys

CXXFILT = "c++filt"


def parse(paths):
    rows, cur = {}, None
    for p in paths:
        for ln in open(p, errors="replace"):
            m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis=kernel-resource-usage\]", ln)
            if not m:
                continue
            s = m.group(1).strip()
            if s.startswith("Function Name:"):
                cur = s.split(":", 1)[1].strip()
                rows[cur] = {}
            elif cur and ":" in s:
                k, v = s.rsplit(":", 1)
                rows[cur][k.strip()] = v.strip()
    return rows


def demangle(names):
    try:
        out = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:       # noqa: BLE001
        return {n: n for n in names}


def main():
    rows = parse(sys.argv[1:])
    names = demangle(list(rows))
    lines = []
    for n, r in rows.items():
        d = re.sub(r"^void ", "", names[n])
        d = re.sub(r"\(.*", "", d)
        lines.append(f"{d[:64]:64s} VGPR {r.get('VGPRs'):>3} SGPR {r.get('TotalSGPRs'):>3} spill v{r.get('VGPRs Spill'):>3} s{r.get('SGPRs Spill'):>3} "
                     f"scratch {r.get('ScratchSize [bytes/lane]'):>4} occ {r.get('Occupancy [waves/SIMD]')} lds {r.get('LDS Size [bytes/block]')}")
    print("\n".join(sorted(lines)))


if __name__ == "__main__":
    main()
