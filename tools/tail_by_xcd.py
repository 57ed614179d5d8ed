"""When the waves of a march launch leave, per XCD (wave-timing profile, raw slots): is the end of a launch uneven between
the XCDs' queues?      python tools/tail_by_xcd.py [dots]"""
import ctypes
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from photon_amd import scenes  # noqa: E402
from photon_amd.library import PhotonLibrary  # noqa: E402

dots = int(sys.argv[1]) if len(sys.argv) > 1 else 25
lib = PhotonLibrary()
work = os.path.join(tempfile.gettempdir(), "photon_bench")
os.makedirs(work, exist_ok=True)
path = scenes.config("C3", work).density_grad_filename
call = scenes.bos_scene(n_dots=dots, density_grad_filename=path)
scene = lib.scene_create(call)
vol = lib.volume_load_nrrd(path, 2)
H, W = call.image_shape
img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
for _ in range(20):
    scene.trace(img.data_ptr(), vol, 2)
scene.set_march_profile(True)
scene.stats_begin()
for _ in range(8):
    scene.trace(img.data_ptr(), vol, 2)
st = scene.stats_end()
raw = (ctypes.c_ulonglong * (64 * 8))()
f = lib.lib.photon_scene_march_profile_raw
f.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
M = (1 << 64) - 1
acc = np.zeros((8, 4))
for launch in range(8):
    assert f(scene.handle, launch, raw) == 0
    a = np.array(list(raw), dtype=object).reshape(64, 8)
    t0 = min(M - int(r[0]) for r in a if r[0])
    for x in range(8):
        rows = [r for i, r in enumerate(a) if i % 8 == x and r[7]]
        waves = sum(int(r[7]) for r in rows)
        end_min = min(M - int(r[4]) for r in rows) - t0
        end_max = max(int(r[6]) for r in rows) - t0
        end_mean = (sum(int(r[5]) for r in rows) - waves * t0) / waves
        acc[x] += [waves, end_min * 1e-5, end_mean * 1e-5, end_max * 1e-5]
acc /= 8
print(f"dots {dots}: march {st.march_ms / 8:.3f} ms per launch; per XCD (mean of 8 launches): waves, first exit, mean exit, last exit [ms]")
for x in range(8):
    print(f"  XCD slot {x}: {acc[x, 0]:6.0f}  {acc[x, 1]:7.3f}  {acc[x, 2]:7.3f}  {acc[x, 3]:7.3f}")
