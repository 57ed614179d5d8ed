"""Build libparallel_ray_tracing.so (HIP, gfx950) in-tree with hipcc.

    python -m photon_amd.build [--force]

The output lands next to this file (photon_amd/libparallel_ray_tracing.so): git-ignored, but it
travels with the tree to the GPU box.  hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_NAME = "libparallel_ray_tracing.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)

SOURCES = [os.path.join(CSRC, "photon_core.hip")]
HEADERS = [os.path.join(CSRC, h) for h in ("device_vec.hpp", "device_volume.hpp", "device_volume_coop.hpp", "device_volume_extra.hpp", "device_optics.hpp")] + [
    os.path.join(ROOT, "include", "parallel_ray_tracing.h"),
    os.path.join(ROOT, "include", "photon_det_math.h"),
    os.path.join(ROOT, "include", "photon_philox.h"),
]

# -ffp-contract=off: fused multiply-adds only where the source says fmaf()/fma() -- the rounding
#   sequence of a ray is part of the parity contract (include/photon_det_math.h).
# -munsafe-fp-atomics: atomicAdd(float*) is one global_atomic_add_f32, never a CAS loop.
# f32 divide / sqrt stay correctly rounded (hipcc default; made explicit).
# -fno-slp-vectorize: v_pk_*_f32 issue at half the rate of the unpacked forms on gfx950 (measured:
#   tools/ubench/fma_rate.hip), so SLP packing only adds operand shuffles.
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
    "-munsafe-fp-atomics", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function",
]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS + [os.path.abspath(__file__)])


def build_library(force: bool = False, verbose: bool = True, extra_flags=()) -> str:
    if not force and not needs_build():
        return LIB_PATH
    cmd = [hipcc_path()] + HIPCC_FLAGS + list(extra_flags) + ["-o", LIB_PATH] + SOURCES
    if verbose:
        print(" ".join(cmd), file=sys.stderr, flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=sys.stderr)
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB_PATH)
