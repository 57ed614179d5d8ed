import sys, os, tempfile, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from photon_amd import scenes
from photon_amd.library import PhotonLibrary
lib = PhotonLibrary()
work = os.path.join(tempfile.gettempdir(), "photon_bench"); os.makedirs(work, exist_ok=True)
t0 = time.perf_counter(); call = scenes.config("C5", work, scale=float(sys.argv[1]) if len(sys.argv) > 1 else 1.0); t1 = time.perf_counter()
scene = lib.scene_create(call)
vol = lib.volume_load_nrrd(call.density_grad_filename, 2)
H, W = call.image_shape
img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
t2 = time.perf_counter(); st = scene.trace(img.data_ptr(), vol, 2, want_stats=True); t3 = time.perf_counter()
st = scene.trace(img.data_ptr(), vol, 2, want_stats=True)
print(json.dumps({"rays": call.num_rays, "sources": call.num_sources, "host_scene_s": round(t1 - t0, 2), "first_trace_s": round(t3 - t2, 3),
                  "total_ms": round(st.total_ms, 2), "march_ms": round(st.march_ms, 2), "Mrays_per_s": round(call.num_rays / st.total_ms * 1e-3, 1),
                  "iters_per_ray": round(st.rk_iterations / call.num_rays, 1), "on_sensor": st.rays_on_sensor}))
