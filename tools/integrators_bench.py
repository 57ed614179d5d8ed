import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from photon_amd import scenes
from photon_amd.library import PhotonLibrary
lib = PhotonLibrary()
work = os.path.join(tempfile.gettempdir(), "photon_bench"); os.makedirs(work, exist_ok=True)
call = scenes.config("C3", work)
scene = lib.scene_create(call)
H, W = call.image_shape
img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
for interp in (1, 2):
    vol = lib.volume_load_nrrd(call.density_grad_filename, interp)
    for algo in (1, 2, 3, 4):
        scene.trace(img.data_ptr(), vol, algo, want_stats=True)
        st = scene.trace(img.data_ptr(), vol, algo, want_stats=True)
        print(f"interp {interp} algorithm {algo}: march {st.march_ms:.2f} ms total {st.total_ms:.2f} ms iters/ray {st.rk_iterations / call.num_rays:.1f}", flush=True)
    vol.free()
