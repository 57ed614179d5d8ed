#!/usr/bin/env python3
"""Static report of the march kernels, no GPU needed: registers, spills, scratch, occupancy (the compiler's
kernel-resource-usage remarks) and instruction counts by pipe (from the gfx950 assembly).

    python tools/kernel_report.py [--match march_kernel] [--flags "-DPHOTON_X=1 ..."] [--loops]

--loops prints, per kernel, every innermost-looking backward-branch region with its instruction mix: the RK4 trip is the
largest one.  Counts are STATIC (instructions in the code), a proxy for the PMC counters in profiles/ -- rare paths
(bricks, gather fallback, repairs) sit in the same function and are listed separately when they are their own loops."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from photon_amd import build  # noqa: E402


def classify(op: str) -> str:
    if op.startswith(("v_mfma", "v_smfma")):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_scratch")):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    return "other"


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), text=True, capture_output=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--match", default="march_kernel")
    ap.add_argument("--flags", default="")
    ap.add_argument("--loops", action="store_true")
    ap.add_argument("--keep", default=None, help="write the assembly here")
    ap.add_argument("--unit", default="photon_march_cubic", help="translation unit under photon_amd/csrc (photon_march_cubic, photon_march_linear, "
                                                                "photon_sensor, ...)")
    args = ap.parse_args()
    src = os.path.join(build.CSRC, args.unit + ".hip")
    with tempfile.TemporaryDirectory() as tmp:
        asm = args.keep or os.path.join(tmp, "k.s")
        cmd = [build.hipcc_path()] + build.HIPCC_FLAGS + args.flags.split() + ["--cuda-device-only", "-S", "-Rpass-analysis=kernel-resource-usage", src, "-o", asm]
        r = subprocess.run(cmd, cwd=build.CSRC, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
        res = collections.OrderedDict()
        cur = None
        for ln in r.stderr.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", ln)
            if m:
                cur = m.group(1)
                res[cur] = {}
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", ln)
            if m and cur:
                res[cur][m.group(1).strip()] = m.group(2)
        text = open(asm).read().splitlines()
    # split the assembly into functions
    funcs, name = {}, None
    for ln in text:
        m = re.match(r"^(\w+):\s*(;.*)?$", ln)
        if m and not ln.startswith(".L"):
            name = m.group(1)
            funcs[name] = []
            continue
        if name is not None:
            if ln.strip().startswith(".end_amdhsa_kernel") or ln.strip().startswith(".Lfunc_end"):
                name = None
                continue
            funcs[name].append(ln)
    names = [n for n in res if args.match in n]
    dm = demangle(names)
    for n in names:
        body = funcs.get(n, [])
        mix = collections.Counter()
        labels, instrs = {}, []
        for ln in body:
            s = ln.strip()
            m = re.match(r"^(\.LBB\w+):", s)
            if m:
                labels[m.group(1)] = len(instrs)
                continue
            if not s or s.startswith((";", ".", "//")):
                continue
            op = s.split()[0]
            instrs.append((op, s))
            mix[classify(op)] += 1
        short = re.sub(r"\(.*", "", dm.get(n, n)).replace("void ", "")
        r_ = res[n]
        print(f"{short:32s} VGPRs {r_.get('VGPRs')} spill {r_.get('VGPRs Spill')} (SGPR spill {r_.get('SGPRs Spill')}) scratch {r_.get('ScratchSize')} B "
              f"occupancy {r_.get('Occupancy')} | static: valu {mix['valu']} salu {mix['salu']} lds {mix['lds']} vmem {mix['vmem']} scratch {mix['scratch']} wait {mix['wait']}")
        if args.loops:
            loops = []
            for i, (op, s) in enumerate(instrs):
                if op.startswith(("s_cbranch", "s_branch")):
                    tgt = s.split()[-1]
                    if tgt in labels and labels[tgt] <= i:
                        loops.append((labels[tgt], i))
            for a, b in sorted(loops, key=lambda t: t[0] - t[1])[:6]:
                c = collections.Counter(classify(op) for op, _ in instrs[a:b + 1])
                print(f"    loop [{a:5d},{b:5d}] {b - a + 1:5d} instrs: valu {c['valu']} salu {c['salu']} lds {c['lds']} vmem {c['vmem']} scratch {c['scratch']} wait {c['wait']}")


if __name__ == "__main__":
    main()
