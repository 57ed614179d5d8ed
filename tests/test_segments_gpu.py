"""The segmented march (photon_scene_set_march_segments / PHOTON_MARCH_SEGMENTS; march_kernel.hpp, march_group): every
ray's march cut into S pieces that different waves handle at different times must return the BITS of the whole march.

Forced segment counts on launches far smaller than the chip -- every piece is handed out while the previous piece of its
group is still being marched, so each hand-off between two waves really waits on the flag -- for both integrators, both
samplers, coherent (BOS) and incoherent, unevenly loaded (PIV through a volume, doomed rays skipped) launches; then the
library's own choice on a launch large enough for it."""
import os

import numpy as np
import pytest

from photon_amd import scenes
from test_parity_gpu import IMAGE_TOL, assert_bit_equal, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def volume_file(workdir):
    rho, sp, org = scenes.bos_volume(48)
    return scenes.write_nrrd(os.path.join(workdir, "seg48.nrrd"), rho, sp, org)


def _dumps(photon, call, folder, segments, monkeypatch):
    """start_ray_tracing with the final ray dumps on: per-ray sensor hit and post-march direction, image."""
    monkeypatch.setenv("PHOTON_MARCH_SEGMENTS", segments)
    pdir, ddir = folder / f"pos_{segments.replace(':', '_')}", folder / f"dir_{segments.replace(':', '_')}"
    pdir.mkdir(parents=True)
    ddir.mkdir(parents=True)
    call.lightray_position_save_path, call.lightray_direction_save_path = str(pdir), str(ddir)
    img = photon.render(call)
    return (np.fromfile(pdir / "pos_0000.bin", np.float32), np.fromfile(ddir / "dir_0000.bin", np.float32), img)


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("algorithm", [1, 2])
def test_segmented_march_returns_the_bits_of_the_whole_march(photon, oracle, volume_file, tmp_path, monkeypatch, interp, algorithm):
    """188 groups on a chip that holds 5120 waves: with 2, 5 and 64 forced segments every piece waits for the wave that
    is still marching the previous one.  Per ray, bit for bit: the direction after the march and the sensor hit (the
    reference's own pos_/dir_ dumps, .cu:3570-3611) equal those of the whole march and of the CPU oracle."""
    monkeypatch.setenv("PHOTON_INTERP", interp)
    call = scenes.bos_scene(n_dots=6, points_per_dot=20, rays_per_source=100, density_grad_filename=volume_file,
                            ray_tracing_algorithm=algorithm)
    call.save_lightrays, call.num_lightrays_save = True, call.num_rays
    whole = _dumps(photon, call, tmp_path, "1", monkeypatch)
    assert np.isfinite(whole[0]).any()
    for shape in ("halving", "uniform"):                        # pieces of 1/2, 1/4, ... of the depth (shipped) / of equal length
        monkeypatch.setenv("PHOTON_MARCH_SEGMENT_SHAPE", shape)
        for seg in ("force:2", "force:5", "force:64"):
            got = _dumps(photon, call, tmp_path / shape, seg, monkeypatch)
            assert_bit_equal(got[0], whole[0], f"sensor hits, {seg} {shape}")
            assert_bit_equal(got[1], whole[1], f"directions after the march, {seg} {shape}")
            assert rel_l2(got[2], whole[2]) <= 1e-12        # same increments, f64 sums in another order
    monkeypatch.delenv("PHOTON_MARCH_SEGMENT_SHAPE")
    cdir = tmp_path / "cpu"
    cdir.mkdir()
    call.lightray_position_save_path = call.lightray_direction_save_path = str(cdir)
    ref, _ = oracle.render(call, interpolation=2 if interp == "cubic" else 1)
    assert_bit_equal(whole[1], np.fromfile(cdir / "dir_0000.bin", np.float32), "directions vs oracle")
    assert rel_l2(whole[2], ref) <= IMAGE_TOL


@pytest.mark.parametrize("interp", [1, 2])
def test_segment_counters_and_uneven_load(photon, volume_file, interp):
    """PIV through a volume, lens-major, doomed rays skipped: trivial and real groups mixed, waves arriving at every
    hand-off at different times.  Device-resident path: the statistics of a segmented trace (iterations, samples, rays
    marched and on the sensor) are those of the whole march, the images agree to f64 summation order, and a second
    segmented trace (the flags' launch epoch) does too."""
    import torch
    call = scenes.piv_scene(n_particles=3000, rays_per_source=40, mie=True, polydisperse=True,
                            density_grad_filename=volume_file, field_half_width=3.0e4, sort_by_tile=True)
    scene = photon.scene_create(call)
    vol = photon.volume_load_nrrd(volume_file, interp)
    H, W = call.image_shape
    out = {}
    for seg in (1, 3, 16, 16):
        scene.set_march_segments(seg)
        img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
        st = scene.trace(img.data_ptr(), vol, 2, want_stats=True)
        out.setdefault(seg, []).append((img.cpu().numpy().astype(np.float64), st))
    ref_img, ref = out[1][0]
    assert ref.rk_iterations > 0 and 0 < ref.rays_marched < call.num_rays        # some rays doomed, the rest marched
    for seg in (3, 16):
        for img, st in out[seg]:
            for f in ("rk_iterations", "volume_samples", "rays_marched", "rays_on_sensor", "sensor_taps"):
                assert getattr(st, f) == getattr(ref, f), (seg, f)
            assert rel_l2(img, ref_img) <= 1e-12
    scene.free()
    vol.free()


def test_large_launch_is_segmented_by_default(photon, workdir):
    """2e6 rays = 31250 groups, six chip fills, through 128^3 (tricubic RK4: ~1 ms per group): the library segments on its
    own (its cost model picks 6 pieces here); same counters, same image as whole marches.  (That the launch's drain shrinks is a
    timing statement: tests/test_zz_perf_bounds_gpu.py.)"""
    import torch
    rho, sp, org = scenes.bos_volume(128)
    volume_file = scenes.write_nrrd(os.path.join(workdir, "seg128.nrrd"), rho, sp, org)
    call = scenes.bos_scene(n_dots=40, points_per_dot=100, rays_per_source=500, density_grad_filename=volume_file)
    scene = photon.scene_create(call)
    vol = photon.volume_load_nrrd(volume_file, 2)
    H, W = call.image_shape
    res = {}
    for seg in (1, -1):
        scene.set_march_segments(seg)
        scene.set_march_profile(True)
        img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
        scene.trace(img.data_ptr(), vol, 2)                     # warm-up
        img.zero_()
        st = scene.trace(img.data_ptr(), vol, 2, want_stats=True)
        res[seg] = (img.cpu().numpy().astype(np.float64), st, scene.march_profile())
    for f in ("rk_iterations", "volume_samples", "rays_marched", "rays_on_sensor"):
        assert getattr(res[-1][1], f) == getattr(res[1][1], f), f
    assert rel_l2(res[-1][0], res[1][0]) <= 1e-12
    whole, seg = res[1][2], res[-1][2]
    print("march profile, whole marches:", whole, "\nmarch profile, segments:", seg, "\nmarch ms:", res[1][1].march_ms, res[-1][1].march_ms)
    assert whole["launches"] == 1 and seg["launches"] == 1 and seg["waves"] > 1000      # (the drain itself: test_zz_perf_bounds_gpu.py)
    scene.free()
    vol.free()
