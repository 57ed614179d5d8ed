"""The cache of freed device blocks behind start_ray_tracing (photon_pool.hip, pool_malloc / pool_free): photon's unchanged
Python builds every scene anew per call, so per call the library would hipMalloc and hipFree ~25 blocks, the ray-state
workspace among them.  Blocks are recycled instead; the cache is bounded and can be emptied."""
import os

import numpy as np
import pytest

from photon_amd import scenes

pytestmark = pytest.mark.gpu


def test_block_cache_recycles_and_trims(photon, workdir):
    """Images only: recycled (unzeroed) blocks and an emptied cache change nothing.  The memory figures (nothing new per
    call, the trim gives the workspace back) are resource bounds: tests/test_zz_perf_bounds_gpu.py."""
    def same(x, y):                                              # up to the order of the f64 atomic adds
        x, y = x.astype(np.float64), y.astype(np.float64)
        return np.linalg.norm(x - y) <= 1e-12 * np.linalg.norm(y)

    rho, sp, org = scenes.bos_volume(48)
    path = scenes.write_nrrd(os.path.join(workdir, "pool48.nrrd"), rho, sp, org)
    call = scenes.bos_scene(n_dots=20, points_per_dot=100, rays_per_source=500, density_grad_filename=path)      # 1e6 rays: 32 MB of ray state
    os.environ["PHOTON_INTERP"] = "cubic"
    first = photon.render(call)
    for _ in range(5):
        img = photon.render(call)
    assert first.any() and same(img, first)                      # recycled (unzeroed) blocks change nothing
    photon.lib.photon_trim_caches.restype = None
    photon.lib.photon_trim_caches()
    assert same(photon.render(call), first)                      # and the next call simply allocates again


def test_scene_free_waits_for_work_on_a_non_blocking_stream(photon, workdir):
    """photon_scene_free hands its blocks to the cache, and the next scene of the same shape gets them at once and
    overwrites them with null-stream copies -- which do not wait for a non-blocking stream.  A scene freed right after an
    asynchronous photon_trace on such a stream must therefore wait for it first: trace scene A (sources on the left) on a
    side stream, free it without synchronising, create scene B of the same shape (sources on the right) and render it;
    both images must be what the same scenes give when everything is synchronised."""
    import torch
    rho, sp, org = scenes.bos_volume(48)
    path = scenes.write_nrrd(os.path.join(workdir, "pool48.nrrd"), rho, sp, org)
    vol = photon.volume_load_nrrd(path, 2)
    a = scenes.bos_scene(n_dots=20, points_per_dot=100, rays_per_source=500, density_grad_filename=path, seed=3)
    b = scenes.bos_scene(n_dots=20, points_per_dot=100, rays_per_source=500, density_grad_filename=path, seed=4)
    H, W = a.image_shape

    def render_sync(call):
        sc = photon.scene_create(call)
        img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
        sc.trace(img.data_ptr(), vol, 2)
        torch.cuda.synchronize()
        sc.free()
        return img.cpu().numpy().astype(np.float64)

    want_a, want_b = render_sync(a), render_sync(b)
    side = torch.cuda.Stream()                                   # non-blocking with respect to the null stream
    for _ in range(3):
        img_a = torch.zeros(H * W, dtype=torch.float32, device="cuda")
        img_b = torch.zeros(H * W, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        sa = photon.scene_create(a)
        sa.trace(img_a.data_ptr(), vol, 2, stream=side.cuda_stream)        # asynchronous: returns while the march runs
        sa.free()                                                          # no synchronisation by the caller
        sb = photon.scene_create(b)                                        # same shape: takes A's blocks from the cache
        sb.trace(img_b.data_ptr(), vol, 2)
        torch.cuda.synchronize()
        sb.free()
        for got, want in ((img_a, want_a), (img_b, want_b)):
            g = got.cpu().numpy().astype(np.float64)
            assert np.linalg.norm(g - want) <= 1e-12 * np.linalg.norm(want)
    vol.free()
