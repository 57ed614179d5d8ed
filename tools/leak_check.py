import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from photon_amd import scenes
from photon_amd.library import PhotonLibrary
lib = PhotonLibrary()
work = tempfile.mkdtemp()
rho, sp, org = scenes.bos_volume(48)
path = scenes.write_nrrd(os.path.join(work, "v.nrrd"), rho, sp, org)
call = scenes.bos_scene(n_dots=20, points_per_dot=50, rays_per_source=200, density_grad_filename=path)
piv = scenes.piv_scene(n_particles=2000, rays_per_source=64, mie=True, polydisperse=True, density_grad_filename=path, field_half_width=2.5e4)
def used():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return (t - f) / 2**20
for mode in ("", "0,0,0"):
    if mode: os.environ["PHOTON_DEVICES"] = mode
    else: os.environ.pop("PHOTON_DEVICES", None)
    os.environ["PHOTON_INTERP"] = "cubic"
    lib.render(call); lib.render(piv)
    a = used()
    for _ in range(60):
        lib.render(call); lib.render(piv)
    b = used()
    print(f"PHOTON_DEVICES={mode or '-'}: device memory in use {a:.1f} -> {b:.1f} MiB after 120 calls")
sc = lib.scene_create(call); vol = lib.volume_load_nrrd(path, 2)
img = torch.zeros(1024 * 1024, device="cuda")
a = used()
for _ in range(300): sc.trace(img.data_ptr(), vol, 2, want_stats=True)
b = used()
for _ in range(300): sc.trace(img.data_ptr(), vol, 2, want_stats=True)
c = used()
sc.stats_begin()
for _ in range(300): sc.trace(img.data_ptr(), vol, 2)
sc.stats_end()
d = used()
sc.stats_begin()
for _ in range(300): sc.trace(img.data_ptr(), vol, 2)
sc.stats_end()
print(f"photon_trace x300 with stats: {a:.1f} -> {b:.1f} -> (300 more) {c:.1f} MiB; in a statistics window: -> {d:.1f} -> (300 more) {used():.1f} MiB")
