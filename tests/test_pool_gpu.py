"""The cache of freed device blocks behind start_ray_tracing (photon_pool.hip, pool_malloc / pool_free): photon's unchanged
Python builds every scene anew per call, so per call the library would hipMalloc and hipFree ~25 blocks, the ray-state
workspace among them.  Blocks are recycled instead; the cache is bounded and can be emptied."""
import os

import numpy as np
import pytest

from photon_amd import scenes

pytestmark = pytest.mark.gpu


def test_block_cache_recycles_and_trims(photon, workdir):
    import torch

    def used_mib():
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        return (total - free) / 2 ** 20

    def same(x, y):                                              # up to the order of the f64 atomic adds
        x, y = x.astype(np.float64), y.astype(np.float64)
        return np.linalg.norm(x - y) <= 1e-12 * np.linalg.norm(y)

    rho, sp, org = scenes.bos_volume(48)
    path = scenes.write_nrrd(os.path.join(workdir, "pool48.nrrd"), rho, sp, org)
    call = scenes.bos_scene(n_dots=20, points_per_dot=100, rays_per_source=500, density_grad_filename=path)      # 1e6 rays: 32 MB of ray state
    os.environ["PHOTON_INTERP"] = "cubic"
    first = photon.render(call)
    photon.render(call)
    a = used_mib()
    for _ in range(10):
        img = photon.render(call)
    b = used_mib()
    assert abs(b - a) < 1.0                                      # calls of one shape allocate nothing new
    assert same(img, first)                                      # recycled (unzeroed) blocks change nothing
    photon.lib.photon_trim_caches.restype = None
    photon.lib.photon_trim_caches()
    c = used_mib()
    assert c < b - 30.0                                          # the cached workspace went back to the runtime
    assert same(photon.render(call), first)                      # and the next call simply allocates again
