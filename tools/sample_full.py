"""Wall time of start_ray_tracing on the reference's sample cases at their real size (inputs: tests/golden/abi_*_full.*,
captured from the reference's own driver): PIV 50 000 particles x 10 000 rays, BOS 120 000 sources x 500 rays."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (first: one HIP runtime per process)
from conftest import load_fixture_call  # noqa: E402
from photon_amd.library import PhotonLibrary  # noqa: E402

lib = PhotonLibrary()
cases = sys.argv[1:] or ["piv_full", "bos_full_im1", "bos_full_im2"]        # python tools/sample_full.py [case ...]
for case in cases:
    call = load_fixture_call(case)
    lib.render(call)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        img = lib.render(call)
        best = min(best, time.perf_counter() - t0)
    print(json.dumps({"case": case, "rays": call.num_rays, "sources": call.num_sources, "call_ms": round(best * 1e3, 2),
                      "Mrays_per_s": round(call.num_rays / best * 1e-6, 1), "image_sum": float(img.sum())}))
