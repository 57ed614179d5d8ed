"""March-only timing of every ray_tracing_algorithm (1 Euler, 2 RK4, 3 rk45, 4 Adams-Bashforth) x both samplers.

In a RENDER every ray enters the volume through its z-max face (sources sit above it, rays travel -z), where the
reference's rk45 and adams_bashforth test ray_inside_box before their first step and return the ray untouched
(trace_rays_through_density_gradients.h:397-419, 1293-1330: restated literally) -- there they cost nothing and do nothing.
So the integrators are timed where they do march: N rays entering the 256^3 BOS volume through its z-MIN face, heading +z,
through photon_trace_volume_rays (host arrays in and out; the kernels' own time comes from rocprofv3 --kernel-trace --stats
of this command, kept as profiles/r06_integrators_kernel_stats.csv).  Usage: python tools/integrators_bench.py [n_rays]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (first: one HIP runtime per process)
from photon_amd import scenes  # noqa: E402
from photon_amd.library import PhotonLibrary  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
lib = PhotonLibrary()
work = os.path.join(tempfile.gettempdir(), "photon_bench")
os.makedirs(work, exist_ok=True)
call = scenes.config("C3", work)
rng = np.random.default_rng(3)
for interp in (1, 2):
    vol = lib.volume_load_nrrd(call.density_grad_filename, interp)
    i = vol.info()
    lo, hi = np.array(i.min_bound), np.array(i.max_bound)
    # rays in narrow cones of 500 (what a BOS source sends): neighbouring lanes stay within a texel or two of each other
    n_src = n // 500
    cx, cy = rng.uniform(lo[0] * 0.8, hi[0] * 0.8, n_src), rng.uniform(lo[1] * 0.8, hi[1] * 0.8, n_src)
    pos = np.stack([np.repeat(cx, 500), np.repeat(cy, 500), np.full(n_src * 500, lo[2] - 2000.0)], 1)
    d = np.stack([rng.normal(0, 1e-4, n_src * 500), rng.normal(0, 1e-4, n_src * 500), np.ones(n_src * 500)], 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for algo in (1, 2, 3, 4):
        vol.trace_rays(pos[:1000], d[:1000], algo)
        t0 = time.perf_counter()
        _, _, steps = vol.trace_rays(pos, d, algo)
        dt = time.perf_counter() - t0
        print(f"interp {interp} algorithm {algo}: {pos.shape[0]} rays in cones of 500 from the z-min face, {steps.mean():.1f} steps per ray, "
              f"call {dt * 1e3:.1f} ms (copies included)", flush=True)
    vol.free()
# and in a render: the march of algorithms 3 and 4 is the entry-point move only
scene = lib.scene_create(call)
H, W = call.image_shape
img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
vol = lib.volume_load_nrrd(call.density_grad_filename, 1)
for algo in (1, 2, 3, 4):
    scene.trace(img.data_ptr(), vol, algo, want_stats=True)
    st = scene.trace(img.data_ptr(), vol, algo, want_stats=True)
    print(f"render C3 (1e7 rays, trilinear) algorithm {algo}: march {st.march_ms:.2f} ms, total {st.total_ms:.2f} ms, "
          f"iterations per ray {st.rk_iterations / call.num_rays:.1f}", flush=True)
