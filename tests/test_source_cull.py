"""The bound behind the source cull of the volume-free path (photon_scene.hip, source_misses_sensor) on its own: host
arithmetic of the product library, no GPU -- held against exact float64 ray tracing of EVERY ray of EVERY source it rules
out, over hundreds of random thick-lens (and, one in five, thin-lens) cameras (focal length, f-number, both radii, thickness, index, object distance,
field, cone) including ones the sample data never visits."""
import copy

import numpy as np
import pytest

from exact_lens import call_lens, lens_samples, trace_thick_lens, trace_thin_lens
from photon_amd import scenes
from photon_amd.library import PhotonLibrary


@pytest.fixture(scope="module")
def lib():
    return PhotonLibrary()


def _sensor_half(call):
    cam = call.camera
    # a hit reaches a pixel only within half a pixel beyond the array either side (.cu:1440-1452, 1803-1815)
    return (cam["pixel_pitch"] * (cam["x_pixel_number"] + 1) / 2.0, cam["pixel_pitch"] * (cam["y_pixel_number"] + 1) / 2.0)


def _check(lib, call, px, py, min_margin=1.0):
    off = lib.sources_missing_sensor(call, px, py)
    if off is None:
        return None
    zc, t, R1, R2, n, pitch = call_lens(call)
    S = np.stack([np.asarray(a, np.float32).astype(np.float64) for a in (call.src_x, call.src_y, call.src_z)], 1)
    hx, hy = _sensor_half(call)
    # every ray of every culled source, apertures opened by a micron: none may come within a micron of a pixel
    if off.any():
        if call.elements[0]["element_type"] == "t":
            hits, alive = trace_thin_lens(S[off], px, py, float(call.image_distance), zc,
                                          float(call.elements[0]["element_properties"]["thin_lens_focal_length"]), pitch,
                                          float(call.camera["z_sensor"]), margin=1.0)
        else:
            hits, alive = trace_thick_lens(S[off], px, py, float(call.image_distance), zc, t, R1, R2, n, pitch,
                                           float(call.camera["z_sensor"]), margin=1.0)
        near = alive & (np.abs(hits[..., 0]) <= hx + min_margin) & (np.abs(hits[..., 1]) <= hy + min_margin)
        assert not near.any(), (int(near.sum()), call_lens(call))
    return off, S


def test_sample_piv_geometry_culls_what_misses(lib):
    """photon's sample PIV frame (field 1.5 x the field of view, run_simulation_02.py:956-958): the bound rules out nearly
    every source whose rays all miss -- and none that has a ray on the sensor."""
    call = scenes.piv_scene(n_particles=1500, rays_per_source=3000, mie=False, seed=3)
    px, py = lens_samples(lib, call)
    off, S = _check(lib, call, px, py)
    zc, t, R1, R2, n, pitch = call_lens(call)
    hits, alive = trace_thick_lens(S, px, py, float(call.image_distance), zc, t, R1, R2, n, pitch, 0.0)
    hx, hy = _sensor_half(call)
    on = alive & (np.abs(hits[..., 0]) <= hx) & (np.abs(hits[..., 1]) <= hy)
    truly_off = ~on.any(1)
    assert not (off & ~truly_off).any()
    assert 0.45 < off.mean() < truly_off.mean() <= 0.62
    assert off.sum() >= 0.9 * truly_off.sum()                           # not lazy: nine in ten of the sources that miss are caught


@pytest.mark.parametrize("seed", range(6))
def test_random_cameras(lib, seed):
    rng = np.random.default_rng(1000 + seed)
    applied = culled = 0
    for trial in range(60):
        f = 10 ** rng.uniform(4.3, 5.5)
        fnum = rng.uniform(1.8, 22.0)
        obj = f * rng.uniform(2.0, 12.0)
        call = scenes.piv_scene(n_particles=48, rays_per_source=1200, mie=False, seed=seed * 100 + trial,
                                ray_cone_pitch_ratio=float(rng.choice([0.4, 0.7, 1.0])))
        from photon_amd.ray_tracing import single_lens_camera
        try:
            geom = single_lens_camera(f, fnum, obj, f * rng.uniform(0.6, 3.0))
        except (ValueError, FloatingPointError):
            continue
        if not np.isfinite(geom["z_lens"]) or not np.isfinite(geom["refractive_index"]):
            continue
        e = copy.deepcopy(geom["element"])
        g = e["element_geometry"]
        if trial % 3 == 0:                                              # an asymmetric, thicker lens of another glass
            g["front_surface_radius"] *= rng.uniform(0.7, 1.6)
            g["back_surface_radius"] *= rng.uniform(0.7, 1.6)
            g["vertex_distance"] += rng.uniform(0, 0.05 * f)
            e["element_properties"]["refractive_index"] = float(rng.uniform(1.3, 2.0))
        if trial % 5 == 4:                                              # every fifth camera carries photon's thin lens instead
            e["element_type"] = "t"
            e["element_properties"]["thin_lens_focal_length"] = float(f * rng.uniform(0.8, 1.3))
        call.elements = [e]
        call.lens_pitch, call.image_distance = geom["lens_pitch"], geom["image_distance"]
        call.element_center, call.element_plane_parameters = geom["element_center"], geom["element_plane_parameters"]
        half_field = obj * rng.uniform(0.02, 0.6)
        n_src = call.src_x.size
        call.src_x = rng.uniform(-half_field, half_field, n_src) + rng.uniform(-0.2, 0.2) * half_field
        call.src_y = rng.uniform(-half_field, half_field, n_src)
        call.src_z = geom["z_object"] + rng.uniform(-0.3, 0.3, n_src) * obj * (trial % 2)      # every other field is DEEP
        call.camera = scenes.sample_camera(bool(trial % 2), n_pixels=int(rng.choice([256, 1024, 2048])),
                                           pixel_pitch=float(rng.uniform(3.0, 30.0)))
        px, py = lens_samples(lib, call)
        res = _check(lib, call, px, py)
        if res is None:
            continue
        applied += 1
        culled += int(res[0].sum())
    assert applied >= 30 and culled > 200, (applied, culled)


def test_geometries_the_bound_does_not_cover_keep_everything(lib):
    base = scenes.piv_scene(n_particles=64, rays_per_source=500, mie=False, seed=1)
    px, py = lens_samples(lib, base)
    assert lib.sources_missing_sensor(base, px, py).any()

    def variant(edit):
        c = copy.deepcopy(base)
        edit(c)
        return lib.sources_missing_sensor(c, px, py)

    def thin_off_plane(c):                                              # a thin lens whose centre is not on its plane
        c.elements[0]["element_type"] = "t"
        c.element_center = np.array([[0.0, 0.0, c.element_center[0][2] + 5.0]])
    def tilted(c): c.element_plane_parameters = np.array([[0.02, 0.0, 1.0, c.element_plane_parameters[0][3]]])
    def flipped(c): c.element_plane_parameters = -np.asarray(c.element_plane_parameters)
    def off_axis(c): c.element_center = np.array([[50.0, 0.0, c.element_center[0][2]]])
    def concave(c): c.elements[0]["element_geometry"]["back_surface_radius"] = 2.0e5
    def caps_meet(c): c.elements[0]["element_geometry"]["vertex_distance"] *= 0.5       # thinner than its two sags: the caps cross inside the aperture

    def twice(c):                                                       # element 0 applied twice by the reference's element path
        c.elements = [c.elements[0], copy.deepcopy(c.elements[0])]
        c.element_center = np.repeat(np.asarray(c.element_center), 2, 0)
        c.element_plane_parameters = np.repeat(np.asarray(c.element_plane_parameters), 2, 0)
        c.element_system_index = np.array([2, 1], np.int32)
    for edit in (thin_off_plane, tilted, flipped, off_axis, concave, caps_meet, twice):
        assert variant(edit) is None, edit.__name__
    c = copy.deepcopy(base)                                              # photon's thin lens in place of the thick one: covered
    c.elements[0]["element_type"] = "t"
    off_thin = lib.sources_missing_sensor(c, px, py)
    assert off_thin is not None and 0.3 < off_thin.mean() < 0.7
    # sources at or below the lens are kept one by one
    c = copy.deepcopy(base)
    c.src_z = np.asarray(c.src_z).copy()
    c.src_z[:8] = float(c.element_center[0][2]) - 10.0
    off = lib.sources_missing_sensor(c, px, py)
    assert not off[:8].any()
