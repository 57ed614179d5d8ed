"""Wall-clock and memory bounds -- everything in the GPU tier that asserts a TIME or a RESOURCE figure rather than a
result.  The file name sorts after every parity file, so under `pytest -x` a noisy box can only stop the run here, after
the oracle comparisons have all been made.  Each timing bound is a median of five alternating A/B measurements and gets
one retry; the exact figures of the round live in profiles/, these are guards."""
import os
import time

import numpy as np
import pytest

from photon_amd import scenes

pytestmark = pytest.mark.gpu


def _ab_medians(run_a, run_b, reps=5):
    """A/B/A/B...: clock or power drift between two measurement blocks lands on both sides."""
    run_a(), run_b()                                        # warm-up: volume cached, blocks in the cache
    ta, tb = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); run_a(); ta.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter(); run_b(); tb.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ta)), float(np.median(tb))


def _with_one_retry(check):
    first = check()
    if first is None:
        return
    second = check()
    assert second is None, (first, second)


@pytest.mark.parametrize("interp", ["cubic", "linear"])
def test_eight_shards_on_one_gpu_cost(photon, workdir, monkeypatch, interp):
    """C3 (1e7 rays, 256^3) through start_ray_tracing as ONE call and as EIGHT shards on one device.  What the
    comparison can hold on ONE GPU: the eight shards are eight marches of an eighth each, and an eighth costs more than an
    eighth of the whole (bench.py share_of_whole: 0.95 tricubic, 0.92 trilinear) -- 5 % and 8 % that no host path can win
    back here and that eight GPUs do not pay one after the other.  Measured 1.02-1.05 (tricubic) and 1.08-1.10 (trilinear)
    of the single call in round 5; the guard is 1.15 / 1.20 + 0.5 ms.  (The images: tests/test_devices_gpu.py.)"""
    monkeypatch.setenv("PHOTON_INTERP", interp)
    call = scenes.config("C3", workdir)

    def one():
        os.environ.pop("PHOTON_DEVICES", None)
        photon.render(call)

    def many():
        os.environ["PHOTON_DEVICES"] = "0,0,0,0,0,0,0,0"
        photon.render(call)

    def check():
        one_ms, many_ms = _ab_medians(one, many)
        print(f"eight shards on one GPU, {interp}: single {one_ms:.2f} ms, sharded {many_ms:.2f} ms, x{many_ms / one_ms:.3f}")
        return None if many_ms <= (1.15 if interp == "cubic" else 1.20) * one_ms + 0.5 else (one_ms, many_ms)
    try:
        _with_one_retry(check)
    finally:
        os.environ.pop("PHOTON_DEVICES", None)


def test_segmented_launch_drains_faster(photon, workdir):
    """2e6 rays through 128^3, tricubic RK4: the library's own segmentation shrinks the launch's drain -- the average time a
    wave slot stands empty at its end (counters and image of the same pair: tests/test_segments_gpu.py)."""
    import torch
    rho, sp, org = scenes.bos_volume(128)
    volume_file = scenes.write_nrrd(os.path.join(workdir, "seg128.nrrd"), rho, sp, org)
    call = scenes.bos_scene(n_dots=40, points_per_dot=100, rays_per_source=500, density_grad_filename=volume_file)
    scene = photon.scene_create(call)
    vol = photon.volume_load_nrrd(volume_file, 2)
    H, W = call.image_shape
    img = torch.zeros(H * W, dtype=torch.float32, device="cuda")

    def drain(seg):
        scene.set_march_segments(seg)
        scene.set_march_profile(True)
        vals = []
        for _ in range(6):
            img.zero_()
            scene.trace(img.data_ptr(), vol, 2, want_stats=True)
            vals.append(scene.march_profile()["drain_ms"])
        return float(np.median(vals[1:]))

    def check():
        whole, seg = drain(1), drain(-1)
        print(f"drain: whole marches {whole:.3f} ms, segmented {seg:.3f} ms")
        return None if seg < whole else (whole, seg)
    try:
        _with_one_retry(check)
    finally:
        scene.free()
        vol.free()


def test_block_cache_memory(photon, workdir):
    """The cache of freed device blocks (photon_pool.hip): calls of one shape allocate nothing new, and
    photon_trim_caches hands the cached ray-state workspace back to the runtime."""
    import torch

    def used_mib():
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        return (total - free) / 2 ** 20

    rho, sp, org = scenes.bos_volume(48)
    path = scenes.write_nrrd(os.path.join(workdir, "pool48.nrrd"), rho, sp, org)
    call = scenes.bos_scene(n_dots=20, points_per_dot=100, rays_per_source=500, density_grad_filename=path)      # 1e6 rays: 32 MB of ray state
    os.environ["PHOTON_INTERP"] = "cubic"
    photon.render(call)
    photon.render(call)
    a = used_mib()
    for _ in range(10):
        photon.render(call)
    b = used_mib()
    assert abs(b - a) < 1.0
    photon.lib.photon_trim_caches.restype = None
    photon.lib.photon_trim_caches()
    c = used_mib()
    assert c < b - 30.0
