#!/usr/bin/env python3
"""Build the library with extra compile-time switches as build/variants/lib_<name>.so, for same-box A/Bs with tools/ab.sh
(which loads a variant through PHOTON_LIBRARY; photon_version() of such a build names its switches).

    python tools/build_variant.py noprio -DPHOTON_PRIO_BASE=0 -DPHOTON_PRIO_TAPS_DPP=0 -DPHOTON_PRIO_BRICK=0
    python tools/build_variant.py slabs2 -DPHOTON_DPP_SLABS=2
    gpurun -- 'tools/ab.sh "--no-other-configs --no-profile --steps 20" default noprio slabs2 default noprio slabs2'

The tuning switches and what was measured for each value: device_volume_coop.hpp (PHOTON_PRIO_*, PHOTON_DPP_SLABS, tile
layers), march_args.hpp (waves per SIMD, pieces), profiles/r05_*_priority*.txt, r05_e_register_slabs.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from photon_amd import build  # noqa: E402


def main():
    if len(sys.argv) < 3 or not all(a.startswith("-D") for a in sys.argv[2:]):
        print(__doc__)
        return 2
    out = os.path.join(ROOT, "build", "variants", f"lib_{sys.argv[1]}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    print(build.build_library(verbose=False, extra_flags=tuple(sys.argv[2:]), out_path=out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
