#!/usr/bin/env python3
"""Drop-in check with the REAL caller (this container only; needs /root/reference): photon's unmodified Python driver
(`run_simulation_02.py` -> `perform_ray_tracing_03.py`) runs the shipped sample BOS case and loads the library the way it
always does -- `ctypes.CDLL(os.path.abspath('../cuda_codes/Debug/libparallel_ray_tracing.so'))` relative to its cwd
(perform_ray_tracing_03.py:1888) -- from a directory layout in which that path is a symlink to THIS repo's
photon_amd/libparallel_ray_tracing.so, exactly the recipe of INTEGRATION.md.  Nothing of the reference is patched
except the py2 / numpy-1 names it uses and the sample's ray counts (shrunk for speed).

What this proves: the unchanged caller finds the file, resolves `start_ray_tracing`, marshals its 29 arguments through
its own argtypes, and the call returns to it.  Without a GPU (this container) the library reports the HIP error on
stderr and leaves the image untouched, as the header promises; on a GPU box the same call renders.

Prints one JSON line.  Usage: python tests/golden/dropin_check.py"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
np.NAN = np.nan
np.infty = np.inf
np.object = object


def main():
    sys.path.insert(0, os.path.join(REF, "python_codes"))
    import scipy.io as sio
    import helper_functions
    import run_simulation_02 as rs
    import perform_ray_tracing_03 as prt
    prt.long = int
    lib = os.path.join(ROOT, "photon_amd", "libparallel_ray_tracing.so")
    assert os.path.exists(lib), "build the library first: python -m photon_amd.build"
    work = tempfile.mkdtemp(prefix="photon_dropin_")
    os.makedirs(os.path.join(work, "python_codes"))
    os.makedirs(os.path.join(work, "cuda_codes", "Debug"))
    os.symlink(lib, os.path.join(work, "cuda_codes", "Debug", "libparallel_ray_tracing.so"))
    os.symlink(os.path.join(REF, "sample-data"), os.path.join(work, "sample-data"))
    os.chdir(os.path.join(work, "python_codes"))
    os.environ.setdefault("LD_LIBRARY_PATH", "")
    calls = []
    real_cdll = prt.ctypes.CDLL

    def watching_cdll(path, *a, **k):           # observes what the reference loads; loads exactly that
        h = real_cdll(path, *a, **k)
        calls.append({"path": path, "resolves_to": os.path.realpath(path), "has_symbol": hasattr(h, "start_ray_tracing")})
        return h

    prt.ctypes.CDLL = watching_cdll
    m = sio.loadmat(os.path.join(REF, "sample-data", "bos", "parameters", "sample-parameters.mat"),
                    struct_as_record=False, squeeze_me=True)
    p = {k: v for k, v in m.items() if not k.startswith("__")}
    for k in list(p):
        if hasattr(p[k], "_fieldnames"):
            p[k] = helper_functions._todict(p[k])
    p["output_data"]["image_directory"] = os.path.join(work, "images")
    for i in p:
        if isinstance(p[i], (bytes, list, str)):
            continue
        for j in p[i]:
            if type(p[i][j]) is int:
                p[i][j] = float(p[i][j])
    p["bos_pattern"]["grid_point_number"] = 30
    p["bos_pattern"]["particle_number_per_grid_point"] = 20
    p["bos_pattern"]["lightray_number_per_particle"] = 50.0
    images = []
    orig = rs.perform_ray_tracing_03

    def wrapped(*a, **k):
        I, I_raw = orig(*a, **k)
        images.append(float(np.asarray(I_raw, dtype=np.float64).sum()))
        return I, I_raw

    rs.perform_ray_tracing_03 = wrapped
    sys.stdout.flush()
    rs.run_simulation_02(p)
    os.chdir(ROOT)
    shutil.rmtree(work, ignore_errors=True)
    print("DROPIN " + json.dumps({"cdll_calls": calls, "images_returned": len(images), "raw_image_sums": images}))


if __name__ == "__main__":
    main()
