"""C5 (Mie PIV + density volume, 1e6 polydisperse particles x 40 rays = 4e7 rays, 256^3 tricubic RK4) on ONE GPU:
device-resident trace and the PCIe-inclusive start_ray_tracing call (scene upload, device Morton sort, image in / out).
    python tools/c5_full.py [scale [interpolation: 2 = tricubic (default), 1 = trilinear]]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (first: one HIP runtime per process)
from photon_amd import scenes  # noqa: E402
from photon_amd.library import PhotonLibrary  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
interp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
assert torch.cuda.is_available()
lib = PhotonLibrary()
work = os.path.join(tempfile.gettempdir(), "photon_bench")
os.makedirs(work, exist_ok=True)
t0 = time.perf_counter()
call = scenes.config("C5", work, scale=scale)
t1 = time.perf_counter()
scene = lib.scene_create(call)
vol = lib.volume_load_nrrd(call.density_grad_filename, interp)
H, W = call.image_shape
img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
t2 = time.perf_counter()
st = scene.trace(img.data_ptr(), vol, 2, want_stats=True)       # first trace: includes the device Morton sort of the sources
t3 = time.perf_counter()
st = scene.trace(img.data_ptr(), vol, 2, want_stats=True)
os.environ["PHOTON_INTERP"] = "cubic" if interp == 2 else "linear"
lib.render(call)                                                # pays the NRRD parse + volume build
t4 = time.perf_counter()
lib.render(call)                                                # what photon sees per frame: scene upload + sort + trace + image out
t5 = time.perf_counter()
print(json.dumps({"rays": call.num_rays, "sources": call.num_sources, "host_scene_s": round(t1 - t0, 2),
                  "first_trace_s": round(t3 - t2, 3), "total_ms": round(st.total_ms, 2), "march_ms": round(st.march_ms, 2),
                  "Mrays_per_s": round(call.num_rays / st.total_ms * 1e-3, 1), "rays_marched": st.rays_marched,
                  "iters_per_ray": round(st.rk_iterations / call.num_rays, 1), "on_sensor": st.rays_on_sensor,
                  "abi_call_ms": round((t5 - t4) * 1e3, 1), "abi_call_Mrays_per_s": round(call.num_rays / (t5 - t4) * 1e-6, 1)}))
