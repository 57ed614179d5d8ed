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

UNITS = ("photon_pool", "photon_volume", "photon_scene", "photon_march", "photon_march_linear", "photon_march_cubic", "photon_march_extra",
         "photon_sensor", "photon_trace", "photon_post", "photon_abi", "photon_sort", "photon_version")
SOURCES = [os.path.join(CSRC, u + ".hip") for u in UNITS]
HEADERS = [os.path.join(CSRC, h) for h in ("device_vec.hpp", "device_volume.hpp", "device_volume_coop.hpp", "device_volume_extra.hpp", "device_optics.hpp",
                                            "march_args.hpp", "march_kernel.hpp", "photon_internal.hpp", "photon_pool.hpp", "photon_sort.hpp")] + [
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
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-munsafe-fp-atomics", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function",
]
# RUNPATH: the unversioned ROCm prefix, so the library finds libamdhip64 on a box whose ROCm point release differs
# from the build box's (hipcc's own default is the versioned /opt/rocm-X.Y.Z/lib of the machine it ran on).
OBJ_DIR = os.path.join(ROOT, "build", "obj")


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def rocm_lib_dir() -> str:
    """lib/ of the ROCm installation whose hipcc compiles the library ($ROCM_PATH, else the prefix hipcc sits in,
    symlinks unresolved: /opt/rocm rather than /opt/rocm-7.2.0) -- the RUNPATH and the -L of the link."""
    for prefix in (os.environ.get("ROCM_PATH"), os.path.dirname(os.path.dirname(hipcc_path())), "/opt/rocm"):
        lib = os.path.join(prefix, "lib") if prefix else None
        # a hipcc found as /usr/bin/hipcc names /usr/lib, which exists but holds no HIP runtime: ask for the library itself
        if lib and os.path.exists(os.path.join(lib, "libamdhip64.so")):
            return lib
    return "/opt/rocm/lib"


def link_flags():
    lib = rocm_lib_dir()
    return ["-shared", "-fPIC", "--offload-arch=gfx950", "-no-hip-rt", "-Wl,--enable-new-dtags", f"-Wl,-rpath,{lib}", f"-L{lib}",
            "-lamdhip64", "-lz"]        # zlib: gzip-encoded NRRD payloads (photon_volume.hip, parse_nrrd)


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS + [os.path.abspath(__file__)])


def build_id(extra_flags=()) -> str:
    """What photon_version() reports after the library's name: the commit the tree was built from (with -dirty when the
    library's sources differ from it; 'nogit' on a box without the history) and the non-default -DPHOTON_* switches."""
    try:
        sha = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True, check=True).stdout.strip()
        dirty = subprocess.run(["git", "status", "--porcelain", "--", "photon_amd/csrc", "include", "photon_amd/build.py"], cwd=ROOT,
                               capture_output=True, text=True, check=True).stdout.strip()
        sha += "-dirty" if dirty else ""
    except Exception:       # noqa: BLE001
        sha = "nogit"
    switches = sorted(f for f in extra_flags if f.startswith("-DPHOTON_"))
    return sha + " " + ("default" if not switches else "variant[" + " ".join(f[2:] for f in switches) + "]")


def _stale(src: str, obj: str, key: str) -> bool:
    """An object is reused when it is newer than its source, every header and this script, and was compiled with the same
    command line (recorded next to it)."""
    stamp = obj + ".cmd"
    if not (os.path.exists(obj) and os.path.exists(stamp)):
        return True
    t = os.path.getmtime(obj)
    if any(os.path.getmtime(p) > t for p in [src] + HEADERS + [os.path.abspath(__file__)]):
        return True
    with open(stamp) as f:
        return f.read() != key


def _compile(src: str, extra_flags, verbose: bool, out_dir: str, force: bool) -> str:
    obj = os.path.join(out_dir, os.path.splitext(os.path.basename(src))[0] + ".o")
    flags = list(HIPCC_FLAGS) + list(extra_flags)
    if os.path.basename(src) == "photon_version.hip":
        flags.append('-DPHOTON_BUILD_ID="%s"' % build_id(extra_flags))
    cmd = [hipcc_path()] + flags + ["-c", src, "-o", obj]
    key = " ".join(cmd)
    if not force and not _stale(src, obj, key):
        return obj
    if verbose:
        print(key, file=sys.stderr, flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=sys.stderr)
    with open(obj + ".cmd", "w") as f:
        f.write(key)
    return obj


def build_library(force: bool = False, verbose: bool = True, extra_flags=(), out_path: str = None) -> str:
    """Compile the stale translation units (in parallel) and link the shared library.  Concurrent callers (ranks of one
    node, pytest workers) serialise on a lock file; whoever comes second finds the library fresh.  force=True recompiles
    every unit."""
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    target = out_path or LIB_PATH
    if out_path is None and not force and not needs_build():
        return LIB_PATH
    obj_dir = OBJ_DIR if out_path is None else os.path.join(OBJ_DIR, os.path.basename(target) + ".d")
    os.makedirs(obj_dir, exist_ok=True)
    with open(os.path.join(obj_dir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if out_path is None and not force and not needs_build():
            return LIB_PATH
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), max(1, len(os.sched_getaffinity(0))))) as pool:
            objs = list(pool.map(lambda src: _compile(src, extra_flags, verbose, obj_dir, force), SOURCES))
        tmp = target + f".tmp{os.getpid()}"
        cmd = [hipcc_path()] + link_flags() + ["-o", tmp] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr, flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC, stdout=sys.stderr)
        os.replace(tmp, target)
    return target


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB_PATH)
