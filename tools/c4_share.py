#!/usr/bin/env python3
"""One GPU's share of BASELINE C4: 1.25e7 rays (1/8 of 1e8) through the full 512^3 volume, tricubic RK4."""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from photon_amd import scenes
from photon_amd.library import PhotonLibrary

lib = PhotonLibrary()
work = os.path.join(tempfile.gettempdir(), "photon_bench")
os.makedirs(work, exist_ok=True)
call = scenes.config("C4", work, scale=0.125, volume_n=512)
scene = lib.scene_create(call)
H, W = call.image_shape
for interp in (2, 1):
    vol = lib.volume_load_nrrd(call.density_grad_filename, interp)
    img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    scene.trace(img.data_ptr(), vol, 2, want_stats=True)
    st = scene.trace(img.data_ptr(), vol, 2, want_stats=True)
    print(json.dumps({"config": "C4 share", "sampler": "cubic" if interp == 2 else "linear", "rays": call.num_rays,
                      "march_ms": round(st.march_ms, 2), "total_ms": round(st.total_ms, 2),
                      "Mrays_per_s": round(call.num_rays / st.total_ms * 1e-3, 1),
                      "Gsamples_per_s": round(st.volume_samples / st.march_ms * 1e-6, 1),
                      "iters_per_ray": round(st.rk_iterations / call.num_rays, 1)}), flush=True)
    vol.free()
