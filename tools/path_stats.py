"""Which sampler path the waves of a launch take (debug build of the library with -DPHOTON_PATH_STATS=1,
build/variants/lib_pathstats.so): coherent tile / brick passes / per-lane gather, per wave-sample.
    PHOTON_LIBRARY=build/variants/lib_pathstats.so python tools/path_stats.py [c5 scale | c3] [--linear]"""
import ctypes
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from photon_amd import scenes  # noqa: E402
from photon_amd.library import PhotonLibrary  # noqa: E402

interp = 1 if "--linear" in sys.argv else 2
argv = [a for a in sys.argv if a != "--linear"]
what = argv[1] if len(argv) > 1 else "c5"
scale = float(argv[2]) if len(argv) > 2 else 0.25
lib = PhotonLibrary()
work = os.path.join(tempfile.gettempdir(), "photon_bench")
os.makedirs(work, exist_ok=True)
call = scenes.config("C5", work, scale=scale) if what == "c5" else scenes.config("C3", work)
scene = lib.scene_create(call)
vol = lib.volume_load_nrrd(call.density_grad_filename, interp)
H, W = call.image_shape
img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
scene.trace(img.data_ptr(), vol, 2)
out = (ctypes.c_ulonglong * 8)()
lib.lib.photon_debug_path_stats(out)                      # clear after the warm-up
st = scene.trace(img.data_ptr(), vol, 2, want_stats=True)
lib.lib.photon_debug_path_stats(out)
v = list(out)
ws = v[0] + v[2]
print(json.dumps({"workload": what, "interp": interp, "rays_marched": st.rays_marched, "march_ms": round(st.march_ms, 2), "wave_samples": ws,
                  "coherent_frac": round(v[0] / ws, 4), "tile_fetch_per_coherent": round(v[1] / max(v[0], 1), 3),
                  "brick_passes_per_incoherent": round(v[3] / max(v[2], 1), 3), "brick_fetch_per_pass": round(v[4] / max(v[3], 1), 3),
                  "lanes_per_pass": round(v[6] / max(v[3], 1), 1), "gathered_lanes_per_incoherent": round(v[5] / max(v[2], 1), 3)}))
