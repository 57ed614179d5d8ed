#!/usr/bin/env python3
"""CPU-oracle timings of the small BASELINE configs (C0, C2, a slice of C3) next to the GPU's, for DESIGN.md."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    from oracle_lib import Oracle
    from photon_amd import scenes
    from photon_amd.library import PhotonLibrary
    import bench
    o = Oracle()
    o.set_num_threads(bench.cpu_budget())
    lib = PhotonLibrary()
    work = os.path.join(tempfile.gettempdir(), "photon_bench")
    os.makedirs(work, exist_ok=True)
    for name, call in (("C0", scenes.config("C0")), ("C2", scenes.config("C2")),
                       ("C3 slice (2000 sources)", scenes.bos_scene(n_dots=20, density_grad_filename=scenes.config("C3", work).density_grad_filename))):
        os.environ["PHOTON_INTERP"] = "cubic"
        t0 = time.perf_counter()
        oi, _ = o.render(call, interpolation=2)
        tc = time.perf_counter() - t0
        lib.render(call)
        t0 = time.perf_counter()
        gi = lib.render(call)
        tg = time.perf_counter() - t0
        print(json.dumps({"config": name, "rays": call.num_rays, "cpu_s": round(tc, 3), "cpu_threads": o.num_threads(),
                          "cpu_Mrays_per_s": round(call.num_rays / tc * 1e-6, 4), "gpu_abi_call_s": round(tg, 4),
                          "gpu_Mrays_per_s": round(call.num_rays / tg * 1e-6, 2)}), flush=True)


if __name__ == "__main__":
    main()
