"""Host-side logic without a GPU: marshalling mirror, sensor post-processing, scene helpers,
NRRD round trip, source sharding."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from photon_amd import scenes
from photon_amd.ray_tracing import (postprocess_image, read_tiff_u16, save_images, single_lens_camera,
                                    write_tiff_u16)
from photon_amd.sharding import shard_range


@pytest.mark.parametrize("case", ["piv", "bos_im1", "bos_im2"])
def test_postprocess_matches_reference(case):
    """uint16 image the reference's post-processing (perform_ray_tracing_03.py:2190-2247) made of a
    synthetic raw image == ours, pixel for pixel."""
    a = np.load(os.path.join(GOLDEN, f"postprocess_{case}.npz"))
    with open(os.path.join(GOLDEN, f"abi_{case}.json")) as f:
        pp = json.load(f)["postprocess"]
    shape = tuple(int(v) for v in a["shape"])
    raw = np.zeros(shape[0] * shape[1], np.float32)
    raw[a["raw_index"]] = a["raw_value"]
    out = postprocess_image(raw.reshape(shape), pp["pixel_gain"], int(pp["pixel_bit_depth"]),
                            pp["intensity_rescaling"], pp["image_noise"])
    ref = np.zeros(shape[0] * shape[1], np.uint16)
    ref[a["out_u16_index"]] = a["out_u16_value"]
    assert out.dtype == np.uint16 and np.array_equal(out.reshape(-1), ref)


def test_single_lens_camera_matches_reference_numbers():
    """Values the reference derives for its sample lens (captured via its marshalling code)."""
    g = single_lens_camera(105000.0, 8.0, 700000.0, 100000.0)
    with open(os.path.join(GOLDEN, "abi_bos_im1.json")) as f:
        j = json.load(f)
    a = np.load(os.path.join(GOLDEN, "abi_bos_im1.npz"))
    assert g["lens_pitch"] == j["scalars"]["lens_pitch"]
    assert g["image_distance"] == pytest.approx(j["scalars"]["image_distance"], rel=1e-15)
    eg = j["elements"][0]["element_geometry"]
    assert g["element"]["element_geometry"]["vertex_distance"] == pytest.approx(eg["vertex_distance"], rel=1e-14)
    assert g["element"]["element_properties"]["refractive_index"] == pytest.approx(
        j["elements"][0]["element_properties"]["refractive_index"], rel=1e-14)
    assert np.allclose(g["element_center"], a["element_center"], rtol=1e-14)
    assert np.allclose(g["element_plane_parameters"], a["element_plane_parameters"], rtol=1e-14)
    assert np.float32(g["z_offset"]) == np.float32(j["source"]["z_offset"])
    assert np.float32(g["z_object"]) == a["src_z"][0]      # BOS sources sit on the object plane


def test_nrrd_round_trip(tmp_path):
    rho, sp, org = scenes.bos_volume(16)
    p = scenes.write_nrrd(str(tmp_path / "v.nrrd"), rho, sp, org)
    r2, sp2, org2 = scenes.read_nrrd(p)
    assert np.array_equal(rho, r2) and list(sp2) == list(sp) and list(org2) == list(org)
    # the reference's own sample file parses too
    r, sp, org = scenes.read_nrrd(os.path.join(GOLDEN, "sample-density.nrrd"))
    assert r.shape == (64, 64, 64) and sp == [519.5459, 519.5459, 519.5]
    assert org == [-16365.714, -16365.714, 733634.3]


def test_scene_sizes():
    c3 = scenes.bos_scene()
    assert c3.num_sources == 20000 and c3.num_rays == 10_000_000 and c3.camera["implement_diffraction"]
    c2 = scenes.piv_scene()
    assert c2.num_rays == 1_000_000 and c2.scattering_irradiance.shape == (255, 27)
    c0 = scenes.config("C0")
    assert c0.num_rays == 10_000 and c0.scattering_type == "diffuse"


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 20000, 1_000_003):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_tiff_and_raw_writers_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    raw = rng.gamma(2.0, 3.0, size=(37, 53)).astype(np.float32)
    u16 = postprocess_image(raw, 25.0, 10)
    tif, binf = str(tmp_path / "a.tif"), str(tmp_path / "a.bin")
    save_images(raw, u16, tif, binf)
    assert np.array_equal(read_tiff_u16(tif), u16)
    assert np.array_equal(np.fromfile(binf, np.float32).reshape(raw.shape), raw)
    # the reference's vendored tifffile is not shipped with the tests; check the header by hand
    with open(tif, "rb") as f:
        assert f.read(4) == b"II*\x00"
    write_tiff_u16(str(tmp_path / "odd.tif"), np.arange(15, dtype=np.uint16).reshape(3, 5))
    assert np.array_equal(read_tiff_u16(str(tmp_path / "odd.tif")), np.arange(15).reshape(3, 5))


def test_march_work_queues_partition_every_launch():
    """The work queues of the persistent march (8 XCDs x 4 sub-queues): for any number of 64-ray groups every group
    below it is handed out by exactly one queue, each queue hands its groups out in increasing order (so the first one
    past the end ends the queue), and the consecutive groups of a chunk (16 groups for the tricubic kernels, 128 for the
    trilinear ones) come from ONE queue (L2 locality).  Host restatement of the kernel's own functions, through the C-ABI; no
    GPU needed."""
    import ctypes
    from photon_amd import build
    lib = ctypes.CDLL(build.build_library(verbose=False))
    f = lib.photon_march_queue_group
    f.argtypes = [ctypes.c_uint] * 4
    f.restype = ctypes.c_uint
    size = lib.photon_march_queue_size
    size.argtypes = [ctypes.c_uint] * 4
    size.restype = ctypes.c_uint
    lib.photon_march_queue_count.restype = ctypes.c_uint
    lib.photon_march_queue_chunk.restype = ctypes.c_uint
    subs = lib.photon_march_queue_count() // 8
    assert subs == 4 and lib.photon_march_queue_chunk(2) == 16 and lib.photon_march_queue_chunk(1) == 128
    bad = 0xFFFFFFFF
    assert f(0, 8, 0, 16) == bad and f(0, 0, subs, 16) == bad and size(100, 0, subs, 16) == bad     # out of range: refused, not aliased
    assert f(0, 0, 0, 24) == bad and size(100, 0, 0, 0) == bad                                       # chunks are powers of two
    for C in (16, 128):
        for n_groups in (1, C - 1, C, C + 1, 1023, 8 * C + 5, 32 * C - 1, 32 * C, 64 * C, 64 * C + 1, 19532):
            seen = np.zeros(n_groups, np.int32)
            for x in range(8):
                for sub in range(subs):
                    last, k = -1, 0
                    while True:
                        g = f(k, x, sub, C)
                        assert g > last                    # monotonic: a queue ends at its first group past the launch
                        last = g
                        if g >= n_groups:
                            break
                        seen[g] += 1
                        assert (g // C) % 8 == x and (g // C // 8) % subs == sub
                        k += 1
                    # a segmented launch hands out size x S items per queue (item k = segment k // size of group k % size):
                    # the closed form counts exactly the groups the queue's own enumeration finds below the launch's end
                    assert k == size(n_groups, x, sub, C), (n_groups, x, sub, C)
            assert (seen == 1).all(), (n_groups, C)


def test_march_segments_plan():
    """How many pieces the library cuts a ray's march into (photon_march_segments_plan: pure host arithmetic, the cost model
    of DESIGN.md section 4.1 fitted to tools/segments_sweep.sh).  Pins the plan of the configurations that were measured: the
    measured optima were 4 (equal) / 4-5 (halving) pieces for the C3 job with the tricubic sampler and 12-16 for one GPU's
    eighth of it; with the trilinear sampler (as of the end of round 4) whole marches and 3 pieces; fast kernels and small
    launches march whole."""
    import ctypes
    from photon_amd import build
    lib = ctypes.CDLL(build.build_library(verbose=False))
    plan = lib.photon_march_segments_plan
    plan.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    plan.restype = ctypes.c_int

    def p(rays, depth, algo, interp, cus=256):
        h = ctypes.c_int(-1)
        return plan(rays, depth, algo, interp, cus, ctypes.byref(h)), h.value

    os.environ.pop("PHOTON_MARCH_SEGMENT_SHAPE", None)
    assert p(10_000_000, 256, 2, 2) in ((4, 1), (5, 1))          # C3 headline: 30.5 chip fills -> halving pieces
    assert p(10_000_000, 256, 2, 1) == (1, 0)                    # the same with the trilinear sampler: whole (measured best, 1 < 2 < 3)
    s, h = p(1_250_000, 256, 2, 2)                               # one GPU's eighth: 3.8 fills -> equal pieces, many
    assert h == 0 and 10 <= s <= 16
    s, h = p(1_250_000, 256, 2, 1)
    assert h == 0 and 3 <= s <= 4                                # measured optimum 3
    assert p(10_000_000, 256, 1, 1) == (1, 0)                    # Euler trilinear: a march of 0.2 ms is not worth a hand-off
    assert p(300_000, 256, 2, 2) == (1, 0)                       # less than 1.25 chip fills: whole marches
    assert p(100_000_000 // 2, 512, 2, 2)[0] >= 2                # one launch of C4
    assert p(10_000_000, 8, 2, 2)[0] <= 2                        # the shortest piece is 4 trips
    assert plan(10_000_000, 256, 3, 2, 256, None) == 0           # integrators 3 / 4 are not segmented: refused
