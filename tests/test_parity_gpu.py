"""GPU parity tests: the HIP library (through its C-ABI) against the CPU oracle.

Bars (stated where used):
  * integer / index / table data and every per-ray float quantity that is produced by IEEE
    operations in a fixed order (volume texels, B-spline coefficients, sampler output, marched ray
    position/direction, dumped rays): BIT-EXACT.
  * sensor images: <= 1e-5 relative L2 (BASELINE.json north_star).  The only sources of difference
    are the order of the f32 atomic adds and libm-vs-ocml erf() at 1e-16.
"""
import ctypes
import os

import numpy as np
import pytest

from conftest import load_fixture_call
from photon_amd import scenes

pytestmark = pytest.mark.gpu

IMAGE_TOL = 1e-5


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(a, b, what):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    both_nan = np.isnan(a) & np.isnan(b)
    same = (bits(a) == bits(b)) | both_nan
    assert same.all(), f"{what}: {np.count_nonzero(~same)} of {same.size} values differ; first at " \
                       f"{np.argwhere(~same)[0]}: {a[~same][0]!r} vs {b[~same][0]!r}"


# ------------------------------------------------------------------------------------------------
# host-side tables
# ------------------------------------------------------------------------------------------------
def test_rand_table_matches_glibc(photon, oracle):
    r1, r2 = photon.rand_table(10000)
    o1, o2 = oracle.rand_table(10000)
    assert np.array_equal(r1, o1) and np.array_equal(r2, o2)


# ------------------------------------------------------------------------------------------------
# volume: build, prefilter, samplers, march -- all bit-exact
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def vol_pair(photon, oracle):
    rng = np.random.default_rng(5)
    n = (40, 36, 44)                                  # nz, ny, nx: deliberately unequal
    z, y, x = np.meshgrid(*(np.linspace(-1, 1, k) for k in n), indexing="ij")
    rho = (1.2 + 0.3 * np.exp(-(x ** 2 + 1.5 * y ** 2 + 0.7 * z ** 2) * 3) + 0.01 * rng.standard_normal(n)).astype(np.float32)
    spacing = (310.0, 295.5, 402.25)
    origin = (-6000.0, -5000.0, 400000.0)
    out = {}
    for interp in (1, 2):
        out[interp] = (photon.volume_from_density(rho, spacing, origin, interp),
                       oracle.volume_from_density(rho, spacing, origin, interp))
    yield out
    for g, o in out.values():
        g.free()
        o.free()


@pytest.mark.parametrize("interp", [1, 2])
def test_volume_build_bit_exact(vol_pair, interp):
    g, o = vol_pair[interp]
    gi, oi = g.info(), o.info()
    for f in ("nx", "ny", "nz", "interpolation"):
        assert getattr(gi, f) == getattr(oi, f)
    for f in ("step_size", "data_min"):
        assert np.float32(getattr(gi, f)).view(np.uint32) == np.float32(getattr(oi, f)).view(np.uint32), f
    assert list(gi.min_bound) == list(oi.min_bound) and list(gi.max_bound) == list(oi.max_bound)
    assert_bit_equal(g.download(False), o.download(False), "gradient texels")
    if interp == 2:
        assert_bit_equal(g.download(True), o.download(True), "B-spline coefficients")


def test_sample_nrrd_volume_bit_exact(photon, oracle, golden_dir):
    path = os.path.join(golden_dir, "sample-density.nrrd")
    for interp in (1, 2):
        g, o = photon.volume_load_nrrd(path, interp), oracle.volume_load_nrrd(path, interp)
        assert_bit_equal(g.download(interp == 2), o.download(interp == 2), f"sample volume interp {interp}")
        assert g.info().data_min == o.info().data_min
        g.free()
        o.free()


@pytest.mark.parametrize("interp", [1, 2])
def test_sampler_bit_exact(vol_pair, interp):
    g, o = vol_pair[interp]
    i = g.info()
    rng = np.random.default_rng(11)
    n = 50000
    coords = np.stack([rng.uniform(-2, i.nx + 2, n), rng.uniform(-2, i.ny + 2, n), rng.uniform(-2, i.nz + 2, n)], 1)
    coords[:64] = np.round(coords[:64])              # texel centres / edges
    coords[64:128] = np.floor(coords[64:128]) + 0.5
    assert_bit_equal(g.sample(coords), o.sample(coords), f"sampler interp {interp}")


def test_tricubic_sampler_agrees_with_f64_evaluation(photon):
    """The product's tricubic sampler (photon_volume_sample through the C-ABI) against a float64 evaluation of the
    64-tap B-spline sum that shares no code with kernels or oracle (tests/test_oracle_golden.py: textbook weights,
    einsum): the anchor of the evaluation order both sides define (slab order, Horner weights).  Tolerance: 8 f32
    roundings of the largest tap; a wrong weight, tap or clamp shows at 1e-2 .. 1."""
    from test_oracle_golden import tricubic_anchor_case, tricubic_f64
    rho, coords = tricubic_anchor_case()
    v = photon.volume_from_density(rho, (100.0, 100.0, 100.0), (0.0, 0.0, 750e3), 2)
    coeffs = v.download(True)
    got = v.sample(coords).astype(np.float64)
    want = tricubic_f64(coeffs, coords)
    err = np.abs(got - want) / np.abs(coeffs.reshape(-1, 4)).max(axis=0)
    assert err.max() < 8 * 2.0 ** -24, err.max()
    v.free()


@pytest.mark.parametrize("algorithm", [1, 2])
@pytest.mark.parametrize("interp", [1, 2])
def test_march_bit_exact(vol_pair, interp, algorithm):
    g, o = vol_pair[interp]
    i = g.info()
    rng = np.random.default_rng(3)
    n = 6000
    lo, hi = np.array(i.min_bound), np.array(i.max_bound)
    pos = np.stack([rng.uniform(lo[a] - 0.1 * (hi[a] - lo[a]), hi[a] + 0.1 * (hi[a] - lo[a]), n) for a in range(3)], 1)
    pos[:, 2] = hi[2] + 5000.0                       # start above the volume, head down (-z) ...
    d = np.stack([rng.normal(0, 0.05, n), rng.normal(0, 0.05, n), -np.ones(n)], 1)
    pos[:500] = np.stack([rng.uniform(lo[a], hi[a], 500) for a in range(3)], 1)      # ... some start inside
    d[500:700, 2] = 1.0                              # ... some fly away (miss: must come back untouched)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    gp, gd, gs = g.trace_rays(pos, d, algorithm)
    op, od, os_ = o.trace_rays(pos, d, algorithm)
    assert np.array_equal(gs, os_), "iteration counts differ"
    assert gs.max() > 10
    assert_bit_equal(gp, op, "marched positions")
    assert_bit_equal(gd, od, "marched directions")


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_march_bit_exact_on_adversarial_cases(photon, oracle, seed):
    """Fuzz of the wave-cooperative march against the oracle, bit for bit, on what the hot path meets least often: tiny
    and lopsided grids (down to 4 texels on an axis), a CONSTANT density (every blend sits exactly on data_min: the
    linear branch's below-minimum repair, .h:1056-1065, decides on rounding), rays from every side -- grazing faces and
    edges, axis-parallel (zero direction components: infinities in the slab test), starting on a face, inside, far away,
    pointing away -- packed so that waves are partly coherent, partly not (tile, brick and gather paths in one launch),
    Euler and RK4, both samplers, exact and 8-bit trilinear weights."""
    rng = np.random.default_rng(1000 + seed)
    dims = [(4, 5, 33), (17, 4, 9), (12, 12, 12), (40, 7, 21), (9, 31, 6), (24, 20, 28)][seed]          # nz, ny, nx
    z, y, x = np.meshgrid(*(np.linspace(-1, 1, k) for k in dims), indexing="ij")
    if seed == 2:
        rho = np.full(dims, 1.225, np.float32)                                         # constant: n - 1 == data_min everywhere
    else:
        rho = (1.2 + 0.4 * np.exp(-(x ** 2 + 2 * y ** 2 + 0.5 * z ** 2) * 2) + 0.05 * rng.standard_normal(dims)).astype(np.float32)
    spacing = tuple(float(v) for v in rng.uniform(80.0, 400.0, 3))
    origin = (-1000.0, 500.0, 750e3 + 2000.0)
    n = 4096
    for interp in (1, 2):
        g, o = photon.volume_from_density(rho, spacing, origin, interp), oracle.volume_from_density(rho, spacing, origin, interp)
        i = g.info()
        lo, hi = np.array(i.min_bound, np.float64), np.array(i.max_bound, np.float64)
        ext = hi - lo
        centre = 0.5 * (lo + hi)
        # starts on a sphere around the box, aimed at random points of it; then the special families
        u = rng.standard_normal((n, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        pos = centre + u * 1.5 * np.linalg.norm(ext)
        target = lo + rng.uniform(-0.05, 1.05, (n, 3)) * ext
        d = target - pos
        pos[:400] = lo + rng.uniform(0, 1, (400, 3)) * ext                              # inside starts
        d[:400] = rng.standard_normal((400, 3))
        k = np.arange(400, 800)                                                        # axis-parallel, from outside
        axis = rng.integers(0, 3, k.size)
        d[k] = 0.0
        d[k, axis] = np.where(pos[k, axis] > centre[axis], -1.0, 1.0)
        pos[k] = lo + rng.uniform(0.01, 0.99, (k.size, 3)) * ext
        pos[k, axis] = np.where(d[k, axis] < 0, hi[axis] + 300.0, lo[axis] - 300.0)
        k = np.arange(800, 1000)                                                       # start exactly on the max-z face, heading in
        pos[k] = lo + rng.uniform(0.05, 0.95, (k.size, 3)) * ext
        pos[k, 2] = np.float32(hi[2])
        d[k] = np.stack([rng.normal(0, 0.2, k.size), rng.normal(0, 0.2, k.size), -np.ones(k.size)], 1)
        d[1000:1100] *= -1.0                                                           # pointing away: a miss
        pos[1100:1164] = pos[1100] + rng.uniform(-1e-3, 1e-3, (64, 3))                 # one fully coherent wave's worth
        d[1100:1164] = d[1100]
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        order = rng.permutation(n)
        order[:1164] = np.arange(1164)                                                 # families contiguous, the rest shuffled
        pos, d = pos[order], d[order]
        for bits in ((8, 0) if interp == 1 else (8,)):
            if interp == 1:
                g.set_weight_bits(bits)
                o2 = oracle.volume_from_density(rho, spacing, origin, 1, tex_frac_bits=bits)
            else:
                o2 = o
            for algorithm in (1, 2):
                gp, gd, gs = g.trace_rays(pos, d, algorithm)
                op, od, os_ = o2.trace_rays(pos, d, algorithm)
                what = f"seed {seed} interp {interp} bits {bits} algorithm {algorithm}"
                assert np.array_equal(gs, os_), f"{what}: iteration counts differ at {np.flatnonzero(gs != os_)[:5]}"
                assert_bit_equal(gp, op, what + " positions")
                assert_bit_equal(gd, od, what + " directions")
                # the same rays through the RENDER path's march launch -- persistent waves over the work queues -- whole and
                # cut into 3 and 7 pieces, each piece waiting for the wave that marches the one before (64 groups on a chip
                # of 5120 waves): the resume state (iterations, spins, the repair's last sampled value) must carry the bits
                for segments in (1, 3, 7):
                    qp, qd = g.trace_rays_queued(pos, d, algorithm, segments)
                    assert_bit_equal(qp, op, what + f" positions, queued march in {segments} piece(s)")
                    assert_bit_equal(qd, od, what + f" directions, queued march in {segments} piece(s)")
            if o2 is not o:
                o2.free()
        assert gs.max() > 2
        g.free()
        o.free()


@pytest.mark.parametrize("algorithm", [3, 4, 0, 7])
def test_other_integrators_bit_exact(vol_pair, algorithm):
    """ray_tracing_algorithm 3 (rk45, .h:304-718), 4 (adams_bashforth, .h:1293-1453) and the reference's
    `default: break` (.h:1537).  Both integrators fetch the raw volume trilinearly whatever the sampler;
    rays that enter through a max face come back untouched (both test ray_inside_box before their first
    step), so the test starts rays inside the volume and below it (entering through the z-min face)."""
    for interp in (1, 2):
        g, o = vol_pair[interp]
        i = g.info()
        rng = np.random.default_rng(17 + algorithm)
        n = 3000
        lo, hi = np.array(i.min_bound), np.array(i.max_bound)
        pos = np.stack([rng.uniform(lo[a], hi[a], n) for a in range(3)], 1)            # inside
        d = np.stack([rng.normal(0, 0.08, n), rng.normal(0, 0.08, n), np.ones(n)], 1)
        pos[1000:2000, 2] = lo[2] - 3000.0                                             # below: enter at z-min, head up
        pos[2000:, 2] = hi[2] + 3000.0                                                 # above: enter at z-max, head down
        d[2000:, 2] = -1.0
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        gp, gd, gs = g.trace_rays(pos, d, algorithm)
        op, od, os_ = o.trace_rays(pos, d, algorithm)
        assert np.array_equal(gs, os_), "step counts differ"
        assert_bit_equal(gp, op, f"algorithm {algorithm} positions")
        assert_bit_equal(gd, od, f"algorithm {algorithm} directions")
        if algorithm in (3, 4):
            assert gs[:2000].max() > 10 and (gs[:2000] > 0).mean() > 0.9               # they really march ...
            assert gs[2000:].max() == 0                                                # ... except from a max face
            assert np.array_equal(gd[2000:], d[2000:].astype(np.float32))
        else:
            assert gs.max() == 0 and np.array_equal(gd, d.astype(np.float32))          # moved to the entry point only
            assert np.array_equal(gp[:1000], pos[:1000].astype(np.float32))


# ------------------------------------------------------------------------------------------------
# full pipeline through start_ray_tracing
# ------------------------------------------------------------------------------------------------
def _render_both(photon, oracle, call, interp=1, monkeypatch=None):
    if monkeypatch is not None:
        monkeypatch.setenv("PHOTON_INTERP", "cubic" if interp == 2 else "linear")
    g = photon.render(call)
    o, st = oracle.render(call, interpolation=interp)
    return g, o, st


@pytest.mark.parametrize("case", ["piv", "bos_im1", "bos_im2"])
def test_reference_sample_inputs(photon, oracle, case, monkeypatch):
    """Inputs captured from the reference's own marshalling code for its shipped samples."""
    call = load_fixture_call(case)
    g, o, st = _render_both(photon, oracle, call, 1, monkeypatch)
    assert st.rays_on_sensor > 0 and o.sum() > 0
    assert rel_l2(g, o) <= IMAGE_TOL, rel_l2(g, o)


@pytest.mark.parametrize("case", ["piv", "bos_im1", "bos_im2"])
def test_sample_tiff_pixel_arrays(photon, oracle, case, tmp_path, monkeypatch):
    """The north star's acceptance sentence: the OUTPUT TIFF pixel arrays of the sample PIV and BOS cases.  Both
    raw images go through photon's sensor post-processing (gain, normalise to the brightest pixel, round to the
    sensor's bit depth, stretch to 16 bit; perform_ray_tracing_03.py:2190-2247) and the TIFF writer; the uint16
    arrays read back from the files must agree: a pixel may flip to the neighbouring grey level only where the
    1e-6-level difference of the raw images straddles a rounding boundary."""
    from photon_amd.ray_tracing import postprocess_image, read_tiff_u16, write_tiff_u16
    call = load_fixture_call(case)
    g, o, _ = _render_both(photon, oracle, call, 1, monkeypatch)
    cam = call.camera
    levels = {}
    for tag, raw in (("gpu", g), ("cpu", o)):
        img = postprocess_image(raw, cam["pixel_gain"], cam["pixel_bit_depth"])
        path = write_tiff_u16(str(tmp_path / f"{case}_{tag}.tif"), img)
        levels[tag] = read_tiff_u16(path).astype(np.int64)
    step = 65535 // (2 ** int(cam["pixel_bit_depth"]) - 1)                 # one grey level of the sensor in 16-bit counts
    diff = np.abs(levels["gpu"] - levels["cpu"])
    assert levels["cpu"].max() == 65535 and levels["cpu"].any()
    assert diff.max() <= step + 1, diff.max()
    assert (diff > 0).mean() <= 1e-4, (diff > 0).mean()


def test_c0_plumbing(photon, oracle):
    call = scenes.config("C0")
    g, o, st = _render_both(photon, oracle, call)
    assert o.sum() > 0
    assert rel_l2(g, o) <= IMAGE_TOL


def test_c2_piv_mie_small(photon, oracle):
    call = scenes.piv_scene(n_particles=40, rays_per_source=2500, mie=True)
    g, o, st = _render_both(photon, oracle, call)
    assert st.rays_on_sensor > 1000
    assert rel_l2(g, o) <= IMAGE_TOL, rel_l2(g, o)


def test_piv_polydisperse_mie(photon, oracle):
    call = scenes.piv_scene(n_particles=300, rays_per_source=300, mie=True, polydisperse=True, seed=4)
    g, o, _ = _render_both(photon, oracle, call)
    assert rel_l2(g, o) <= IMAGE_TOL


@pytest.fixture(scope="module")
def small_volume_file(workdir):
    rho, sp, org = scenes.bos_volume(48)
    return scenes.write_nrrd(os.path.join(workdir, "bos48.nrrd"), rho, sp, org)


@pytest.mark.parametrize("interp", [1, 2])
@pytest.mark.parametrize("algorithm", [1, 2])
def test_c3_bos_volume_small(photon, oracle, small_volume_file, interp, algorithm, monkeypatch):
    call = scenes.bos_scene(n_dots=12, points_per_dot=30, rays_per_source=120, density_grad_filename=small_volume_file,
                            ray_tracing_algorithm=algorithm)
    g, o, st = _render_both(photon, oracle, call, interp, monkeypatch)
    assert st.rk_iterations > 40 * call.num_rays          # rays really crossed the 48^3 volume
    assert rel_l2(g, o) <= IMAGE_TOL, rel_l2(g, o)
    # and the volume matters: the undisturbed image differs
    ref = photon.render(scenes.bos_scene(n_dots=12, points_per_dot=30, rays_per_source=120))
    assert rel_l2(g, ref) > 1e-3


def test_c5_piv_with_volume_small(photon, oracle, small_volume_file, monkeypatch):
    call = scenes.piv_scene(n_particles=400, rays_per_source=40, mie=True, polydisperse=True,
                            density_grad_filename=small_volume_file, field_half_width=3.0e4, sort_by_tile=True)
    g, o, st = _render_both(photon, oracle, call, 2, monkeypatch)
    assert st.rk_iterations > 0
    assert rel_l2(g, o) <= IMAGE_TOL


@pytest.mark.parametrize("lens_model,etype", [("thin-lens", "t"), ("apparent", "n"), ("general", "a")])
def test_other_element_types(photon, oracle, lens_model, etype):
    call = scenes.bos_scene(n_dots=8, points_per_dot=20, rays_per_source=64, lens_model=lens_model)
    if etype == "a":                                   # aperture stop: any other element_type char
        call.elements[0]["element_type"] = "a"
    else:
        assert call.elements[0]["element_type"] == etype
    g, o, st = _render_both(photon, oracle, call)
    if etype != "a":
        assert st.rays_on_sensor > 0
    assert rel_l2(g, o) <= IMAGE_TOL if o.any() else not g.any()


def test_single_ray_per_source_is_chief_ray(photon, oracle):
    call = scenes.bos_scene(n_dots=30, points_per_dot=10, rays_per_source=1)
    g, o, _ = _render_both(photon, oracle, call)
    assert rel_l2(g, o) <= IMAGE_TOL


def test_image_is_accumulated_not_overwritten(photon):
    call = scenes.piv_scene(n_particles=20, rays_per_source=200, mie=False)
    first = photon.render(call)
    assert first.any()
    second = photon.render(call, first.copy())          # in/out buffer: renders on top of `first`
    assert rel_l2(second, 2.0 * first.astype(np.float64)) <= 1e-5
    base = np.zeros(call.image_shape, np.float32)
    base[0, 0] = 1000.0                                 # a pixel no ray reaches keeps its value
    third = photon.render(call, base.copy())
    assert third[0, 0] == 1000.0 + first[0, 0]


def test_empty_and_ragged_inputs(photon, oracle):
    # zero sources: nothing happens, image untouched
    call = scenes.piv_scene(n_particles=0, rays_per_source=10, mie=False)
    img = np.full(call.image_shape, 3.0, np.float32)
    assert np.array_equal(photon.render(call, img.copy()), img)
    # ray count that is not a multiple of the wave / workgroup size, more sources than one chunk
    call = scenes.piv_scene(n_particles=137, rays_per_source=77, mie=False, seed=9)
    call.source_point_number = 50
    g, o, _ = _render_both(photon, oracle, call)
    assert rel_l2(g, o) <= IMAGE_TOL


def test_ray_dumps_bit_exact(photon, oracle, small_volume_file, tmp_path):
    call = scenes.bos_scene(n_dots=3, points_per_dot=20, rays_per_source=50, density_grad_filename=small_volume_file)
    call.source_point_number = 25                      # 60 sources -> 3 chunks -> 3 file pairs
    call.save_lightrays = True
    call.num_lightrays_save = 25 * 50
    outs = {}
    for tag, run in (("gpu", lambda c: photon.render(c)), ("cpu", lambda c: oracle.render(c)[0])):
        pdir, ddir = tmp_path / f"{tag}_pos", tmp_path / f"{tag}_dir"
        pdir.mkdir()
        ddir.mkdir()
        call.lightray_position_save_path, call.lightray_direction_save_path = str(pdir), str(ddir)
        run(call)
        outs[tag] = (pdir, ddir)
    for k in range(3):
        for which, prefix in ((0, "pos_"), (1, "dir_")):
            a = np.fromfile(outs["gpu"][which] / f"{prefix}{k:04d}.bin", np.float32)
            b = np.fromfile(outs["cpu"][which] / f"{prefix}{k:04d}.bin", np.float32)
            assert a.size == b.size == 25 * 50 * 3
            assert_bit_equal(a, b, f"{prefix}{k:04d}.bin")
    assert np.isfinite(np.fromfile(outs["gpu"][0] / "pos_0000.bin", np.float32)).any()


@pytest.mark.parametrize("interp", [1, 2])
def test_bos_displacement_matches_the_paraxial_relation(photon, oracle, tmp_path, monkeypatch, interp):
    """End to end on the GPU, through start_ray_tracing: BOS dots rendered without and through a constant-density-gradient
    volume; the centroid shift against photon's own relation displacement = M * Z_D * epsilon / pixel_pitch
    (python_codes/nrrd_functions.py:60-82) -- march, lens and splat together -- and against the oracle's shift."""
    from conftest import bos_displacement_case, image_centroid
    monkeypatch.setenv("PHOTON_INTERP", "cubic" if interp == 2 else "linear")
    c1, c2, predicted = bos_displacement_case(str(tmp_path))
    im1, im2 = photon.render(c1), photon.render(c2)
    (x1, y1), (x2, y2) = image_centroid(im1), image_centroid(im2)
    assert x2 - x1 < 0 and abs(abs(x2 - x1) - predicted) < 0.015 * predicted, (x2 - x1, predicted)
    assert abs(y2 - y1) < 0.01
    o1, _ = oracle.render(c1)
    o2, _ = oracle.render(c2, interpolation=interp)
    assert abs((x2 - x1) - (image_centroid(o2)[0] - image_centroid(o1)[0])) < 1e-4
    assert rel_l2(im2, o2) <= IMAGE_TOL


def test_gpu_ray_dumps_as_the_reference_reader_sees_them(photon, tmp_path, monkeypatch):
    """The GPU library's pos_/dir_/intermediate_*.bin for a BOS image pair, read with our understanding of the wire
    format, against the arrays the REFERENCE's own reader (light_ray_processing.py:74-207) parsed from the oracle's files
    for the same calls (tests/golden/dumps_reference_reader.npz, made in the build container): bit for bit."""
    from conftest import dump_pair_calls
    from test_oracle_golden import check_dumps_against_reference_reader
    monkeypatch.setenv("PHOTON_INTERP", "linear")
    for call in dump_pair_calls(str(tmp_path)):
        photon.render(call)
    check_dumps_against_reference_reader(str(tmp_path))


@pytest.mark.parametrize("algorithm", [1, 2])
def test_intermediate_ray_dumps_bit_exact(photon, oracle, small_volume_file, tmp_path, monkeypatch, algorithm):
    """save_intermediate_ray_data: position / direction at the start of the first N march iterations
    (reference .h:784-790, 1004-1008; files .cu:3613-3670), [ray][slot] float3, NaN = slot not reached."""
    monkeypatch.setenv("PHOTON_INTERP", "linear")
    slots = 12
    call = scenes.bos_scene(n_dots=2, points_per_dot=20, rays_per_source=40, density_grad_filename=small_volume_file,
                            ray_tracing_algorithm=algorithm)
    call.source_point_number = 25                      # 40 sources -> 2 chunks
    call.save_lightrays = True
    call.num_lightrays_save = 25 * 40
    call.save_intermediate_ray_data = True
    call.num_intermediate_positions_save = slots
    outs = {}
    for tag, run in (("gpu", lambda c: photon.render(c)), ("cpu", lambda c: oracle.render(c, interpolation=1)[0])):
        pdir, ddir = tmp_path / f"{tag}_pos", tmp_path / f"{tag}_dir"
        pdir.mkdir()
        ddir.mkdir()
        call.lightray_position_save_path, call.lightray_direction_save_path = str(pdir), str(ddir)
        run(call)
        outs[tag] = (pdir, ddir)
    for k in range(2):
        for which, prefix in ((0, "intermediate_pos_"), (1, "intermediate_dir_")):
            a = np.fromfile(outs["gpu"][which] / f"{prefix}{k:04d}.bin", np.float32)
            b = np.fromfile(outs["cpu"][which] / f"{prefix}{k:04d}.bin", np.float32)
            assert a.size == b.size == 25 * 40 * slots * 3
            assert_bit_equal(a, b, f"{prefix}{k:04d}.bin")
    pos = np.fromfile(outs["gpu"][0] / "intermediate_pos_0000.bin", np.float32).reshape(25 * 40, slots, 3)
    assert np.isfinite(pos[:, 0]).all()                 # slot 0 = entry point on the volume's face
    z = pos[0, :, 2]
    dz = np.diff(z[np.isfinite(z)])
    assert dz.size > 2 and (np.all(dz < 0) or np.all(dz > 0))      # marching monotonically through the volume
    # the cubic branches do not record (as in the reference): files exist, every slot NaN
    monkeypatch.setenv("PHOTON_INTERP", "cubic")
    photon.render(call)
    assert np.isnan(np.fromfile(outs["cpu"][0] / "intermediate_pos_0000.bin", np.float32)).all()


def test_position_noise_hook(photon, oracle, monkeypatch):
    """add_pos_noise: seeded Gaussian jitter of the sensor hit (reference: time-seeded cuRAND, so
    only our two implementations can be compared).  Erf and 4-pixel splat paths."""
    monkeypatch.setenv("PHOTON_NOISE_SEED", "77")
    oracle.set_noise_seed(77)
    for call in (scenes.bos_scene(n_dots=6, points_per_dot=20, rays_per_source=100),
                 scenes.piv_scene(n_particles=60, rays_per_source=400, mie=False, seed=2)):
        clean = photon.render(call)
        call.add_pos_noise, call.pos_noise_std = True, 0.5
        g, o, _ = _render_both(photon, oracle, call)
        assert rel_l2(g, o) <= IMAGE_TOL, rel_l2(g, o)
        assert rel_l2(g, clean) > 1e-2                                  # the noise really blurs the image
    oracle.set_noise_seed(0)


def test_statistics_window_sums_the_traces(photon, small_volume_file):
    """photon_scene_stats_begin / _end: counters and event times summed over the traces of a window (no host sync inside
    it) equal what the same traces report one by one; the march's own clock stamps give a plausible shader clock; per-call
    stats inside an open window are refused."""
    import torch
    from photon_amd.library import PhotonError
    call = scenes.bos_scene(n_dots=6, points_per_dot=20, rays_per_source=100, density_grad_filename=small_volume_file)
    H, W = call.image_shape
    scene = photon.scene_create(call)
    vol = photon.volume_load_nrrd(small_volume_file, 2)
    img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    one = scene.trace(img.data_ptr(), vol, 2, want_stats=True)
    assert one.traces == 1 and one.rays_marched == call.num_rays and one.march_ms > 0
    single = img.cpu().numpy().copy()
    img.zero_()
    scene.stats_begin()
    for _ in range(3):
        scene.trace(img.data_ptr(), vol, 2)
    with pytest.raises(PhotonError):
        scene.trace(img.data_ptr(), vol, 2, want_stats=True)
    st = scene.stats_end()
    assert st.traces == 3 and st.rays_launched == 3 * call.num_rays
    for f in ("rays_on_sensor", "rk_iterations", "volume_samples", "sensor_taps", "rays_marched"):
        assert getattr(st, f) == 3 * getattr(one, f), f
    assert 0 < st.march_ms <= st.total_ms
    assert 1000.0 < st.shader_clock_mhz < 3000.0 and st.march_wave_ms > 0
    assert rel_l2(img.cpu().numpy(), 3 * single) <= 1e-6
    with pytest.raises(PhotonError):
        scene.stats_end()                                   # no window open any more
    scene.free()
    vol.free()


def test_scene_slices_keep_job_wide_noise_ids(photon, small_volume_file):
    """photon_scene_set_source_base: a scene that holds only a slice of the job's sources (one rank of a sharded job)
    draws the noise the whole-job scene draws for those rays -- position noise in the sensor stage, gradient noise in
    the Euler march -- so the sum of the slices is the whole render."""
    import copy
    import torch
    call = scenes.bos_scene(n_dots=4, points_per_dot=15, rays_per_source=64, density_grad_filename=small_volume_file,
                            ray_tracing_algorithm=1)
    H, W = call.image_shape
    vol = photon.volume_load_nrrd(small_volume_file, 1)

    def render(c, base, n):
        sc = photon.scene_create(c)
        sc.set_noise(add_pos_noise=True, pos_noise_std=0.4, add_ngrad_noise=True, ngrad_noise_std=2e-8, seed=31)
        sc.set_source_base(base)
        img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
        sc.trace(img.data_ptr(), vol, 1, 0, n)
        torch.cuda.synchronize()
        sc.free()
        return img.cpu().numpy().astype(np.float64)

    whole = render(call, 0, call.num_sources)
    cut = 23
    parts = np.zeros_like(whole)
    for lo, hi in ((0, cut), (cut, call.num_sources)):
        c = copy.copy(call)
        for f in ("src_x", "src_y", "src_z", "src_radiance", "src_diameter_index"):
            setattr(c, f, getattr(call, f)[lo:hi])
        parts += render(c, lo, hi - lo)
    assert rel_l2(parts, whole) <= 1e-6
    c = copy.copy(call)                                                  # and without the base the second slice draws other numbers
    for f in ("src_x", "src_y", "src_z", "src_radiance", "src_diameter_index"):
        setattr(c, f, getattr(call, f)[cut:])
    assert rel_l2(render(c, 0, call.num_sources - cut) + render(call, 0, cut), whole) > 1e-3
    vol.free()


def test_gradient_noise_hook(photon, oracle, small_volume_file, monkeypatch):
    """add_ngrad_noise: Gaussian noise on dn/dx, dn/dy in the Euler march (trilinear branch)."""
    monkeypatch.setenv("PHOTON_NOISE_SEED", "5")
    monkeypatch.setenv("PHOTON_INTERP", "linear")
    oracle.set_noise_seed(5)
    call = scenes.bos_scene(n_dots=6, points_per_dot=20, rays_per_source=100, density_grad_filename=small_volume_file,
                            ray_tracing_algorithm=1)
    clean = photon.render(call)
    call.add_ngrad_noise, call.ngrad_noise_std = True, 2e-8
    g = photon.render(call)
    o, st = oracle.render(call, interpolation=1)
    assert st.rk_iterations > 0
    assert rel_l2(g, o) <= IMAGE_TOL, rel_l2(g, o)
    assert rel_l2(g, clean) > 1e-3
    oracle.set_noise_seed(0)


@pytest.mark.parametrize("algorithm", [3, 4, 9])
def test_render_with_other_integrators(photon, oracle, small_volume_file, algorithm):
    """The whole pipeline with ray_tracing_algorithm 3 / 4 / out-of-enum (reference: no-op default).  In the
    BOS geometry every ray enters the volume through its z-max face, where rk45 and adams_bashforth stop
    before their first step (as the reference's would): the image equals the no-op one, and the oracle's."""
    call = scenes.bos_scene(n_dots=6, points_per_dot=20, rays_per_source=100, density_grad_filename=small_volume_file,
                            ray_tracing_algorithm=algorithm)
    g = photon.render(call)
    o, _ = oracle.render(call)
    assert g.any() and rel_l2(g, o) <= IMAGE_TOL, rel_l2(g, o)
    call.ray_tracing_algorithm = 0
    assert np.array_equal(photon.render(call), g)


def test_scene_generation_on_device(photon, oracle, monkeypatch, tmp_path):
    """include/parallel_ray_tracing.h section 3: BOS and PIV sources and the synthetic
    Gaussian volume built in HBM are bit-identical to the CPU restatement (and, for BOS, to the numpy arrays the
    host path uploads), and a scene created from them renders the same image as one created from host arrays."""
    import torch
    centres, disc = scenes.bos_pattern(n_dots=9, points_per_dot=31, seed=8)
    rho, sp, org = scenes.bos_volume(40)
    centre = [org[a] + sp[a] * (40 - 1) / 2.0 for a in range(3)]
    call = scenes.bos_scene(n_dots=9, points_per_dot=31, rays_per_source=90, seed=8)
    z_obj = float(call.src_z[0])
    # --- BOS sources
    src = photon.sources_bos(centres, disc, z_obj, 10.0)
    assert src.count() == call.num_sources
    got, want = src.download(), oracle.sources_bos(centres, disc, z_obj, 10.0)
    for key in ("x", "y", "z", "radiance", "diameter_index"):
        assert np.array_equal(got[key], want[key]), key
    assert np.array_equal(got["x"], call.src_x.astype(np.float32)) and np.array_equal(got["y"], call.src_y.astype(np.float32))
    # --- Gaussian volume on the device == CPU restatement, bit for bit
    for interp in (1, 2):
        g = photon.volume_gaussian(40, sp, org, 1.225, 0.2, centre, 8.0e3 * 66300.0 / 66300.0, interp)
        o = oracle.volume_gaussian(40, sp, org, 1.225, 0.2, centre, 8.0e3, interp)
        assert_bit_equal(g.download(interp == 2), o.download(interp == 2), f"gaussian volume interp {interp}")
        assert g.info().data_min == o.info().data_min and g.info().step_size == o.info().step_size
        if interp == 2:
            vol = g
        else:
            g.free()
        o.free()
    # --- render: scene from generated sources + generated volume == scene from host arrays + host density
    vol_host = photon.volume_from_density(rho, sp, org, 2)
    H, W = call.image_shape
    imgs = []
    for scene, v in ((photon.scene_create_from_sources(call, src), vol), (photon.scene_create(call), vol_host)):
        img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
        st = scene.trace(img.data_ptr(), v, 2, want_stats=True)
        assert st.rays_launched == call.num_rays and st.rk_iterations > 0
        imgs.append(img.cpu().numpy())
        scene.free()
    assert imgs[0].any() and rel_l2(imgs[0], imgs[1]) <= IMAGE_TOL          # (volumes differ by the exp's last ulp)
    src.free(); vol.free(); vol_host.free()
    # --- the same field as an NRRD file: read back by our loader (and by scenes.read_nrrd) it is the volume built in place
    path = photon.density_gaussian_write_nrrd(str(tmp_path / "gauss40.nrrd"), 40, sp, org, 1.225, 0.2, centre, 8.0e3)
    from_file, in_place = photon.volume_load_nrrd(path, 1), photon.volume_gaussian(40, sp, org, 1.225, 0.2, centre, 8.0e3, 1)
    assert_bit_equal(from_file.download(), in_place.download(), "NRRD written from the device vs volume built in place")
    rho_file, sp_file, org_file = scenes.read_nrrd(path)
    assert rho_file.shape == (40, 40, 40) and np.allclose(sp_file, sp) and np.allclose(org_file, org)
    np.testing.assert_allclose(rho_file, rho, rtol=3e-7)
    from_file.free(); in_place.free()
    # --- PIV field: bit-identical to the CPU restatement, with and without a diameter distribution
    lo, hi = (-3.0e4, -3.0e4, -7.5e3), (3.0e4, 3.0e4, 7.5e3)
    cdf = np.cumsum(np.full(27, 1.0 / 27.0))
    for c in (None, cdf):
        s = photon.sources_piv(1234, 50_001, lo, hi, z_obj, 730.0, 500.0, c)
        got, want = s.download(), oracle.sources_piv(1234, 50_001, lo, hi, z_obj, 730.0, 500.0, c)
        for key in ("x", "y", "z", "radiance", "diameter_index"):
            assert np.array_equal(got[key], want[key]), key
        s.free()
    # a PIV scene straight from the generator renders like the same sources passed through the host
    pcall = scenes.piv_scene(n_particles=300, rays_per_source=200, mie=True, polydisperse=True, seed=3)
    s = photon.sources_piv(77, 300, lo, hi, z_obj, 730.0, 500.0, cdf)
    d = s.download()
    pcall.src_x, pcall.src_y, pcall.src_z = d["x"], d["y"], d["z"]
    pcall.src_radiance, pcall.src_diameter_index = d["radiance"], d["diameter_index"]
    host_img = photon.render(pcall)
    scene = photon.scene_create_from_sources(pcall, s)
    assert 80 <= scene.live_rays() <= 125             # of 200 lens samples (about half have r1 <= 0.5): the generator's box stands in for the sources it made
    img = torch.zeros(host_img.size, dtype=torch.float32, device="cuda")
    scene.trace(img.data_ptr())
    assert host_img.any() and np.array_equal(img.cpu().numpy().reshape(host_img.shape), host_img)
    o_img, _ = oracle.render(pcall)
    assert rel_l2(host_img, o_img) <= IMAGE_TOL
    scene.free(); s.free()
    # a field wider than the camera sees (photon's sample frame: 1.5 x): the sources that miss the sensor are ruled out on the
    # device from the GENERATED arrays exactly as the host bound rules them out from their download
    from exact_lens import lens_samples
    wide = photon.sources_piv(78, 2000, (-7.5e4, -7.5e4, -7.5e3), (7.5e4, 7.5e4, 7.5e3), z_obj, 730.0, 500.0, cdf)
    d = wide.download()
    wcall = scenes.piv_scene(n_particles=2000, rays_per_source=200, mie=True, polydisperse=True, seed=3)
    wcall.src_x, wcall.src_y, wcall.src_z = d["x"], d["y"], d["z"]
    wcall.src_radiance, wcall.src_diameter_index = d["radiance"], d["diameter_index"]
    scene = photon.scene_create_from_sources(wcall, wide)
    kept = scene.live_sources()
    off = photon.sources_missing_sensor(wcall, *lens_samples(photon, wcall))
    assert kept is not None and np.array_equal(kept, np.flatnonzero(~off)) and 0.3 * 2000 < kept.size < 0.6 * 2000
    img = torch.zeros(host_img.size, dtype=torch.float32, device="cuda")
    scene.trace(img.data_ptr())
    assert np.array_equal(img.cpu().numpy().reshape(host_img.shape), photon.render(wcall))
    scene.free(); wide.free()


@pytest.mark.parametrize("interp", [1, 2])
def test_odd_geometry(photon, oracle, workdir, interp, monkeypatch):
    """The less-trodden corners in one scene: rotated camera (rotation / inverse rotation matrices, .cu:2045-2122),
    non-square sensor, Cauchy dispersion (finite Abbe number, .cu:622-636) with lossy glass (absorbance +
    transmission, .cu:838-853), and a volume with three different sizes and spacings that only part of the rays
    cross."""
    nz, ny, nx = 56, 48, 40
    sp = (310.0, 270.0, 235.0)
    org = (-0.5 * sp[0] * (nx - 1) + 1500.0, -0.5 * sp[1] * (ny - 1), 300000.0)
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    rho = (1.225 + 0.15 * np.exp(-(((x - 17.0) * sp[0]) ** 2 + ((y - 26.0) * sp[1]) ** 2 + ((z - 30.0) * sp[2]) ** 2)
                                 / (2 * 2.5e3 ** 2))).astype(np.float32)
    path = scenes.write_nrrd(os.path.join(workdir, "odd.nrrd"), rho, sp, org)
    call = scenes.bos_scene(n_dots=40, points_per_dot=12, rays_per_source=64, density_grad_filename=path,
                            field_half_width=1.2e4, seed=21)
    ax, ay = np.deg2rad(0.4), np.deg2rad(-0.3)
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    R = Rx @ Ry
    call.camera.update(x_pixel_number=640, y_pixel_number=512, pixel_pitch=21.0, x_camera_angle=float(ax),
                       y_camera_angle=float(ay), rotation_matrix=R, inverse_rotation_matrix=R.T)
    props = call.elements[0]["element_properties"]
    props.update(abbe_number=50.0, transmission_ratio=0.9, absorbance_rate=2.0e-6)
    call.beam_wavelength = 532.0                                  # nm: the Cauchy formula's lambda_D/F/C are (.cu:622)
    assert call.image_shape == (512, 640)
    g, o, st = _render_both(photon, oracle, call, interp, monkeypatch)
    assert 0 < st.rk_iterations < 52 * call.num_rays              # the rotated view clips the volume for part of the rays
    assert st.rays_on_sensor > 0 and o.any()
    assert rel_l2(g, o) <= IMAGE_TOL, rel_l2(g, o)


def test_texture_unit_weights(photon, oracle, small_volume_file, monkeypatch):
    """PHOTON_TEX_WEIGHTS=fixed8 / photon_volume_set_weight_bits(8): trilinear weights rounded to 8 fractional
    bits, the documented arithmetic of the texture unit the reference's tex3D() runs on.  Sampler and marched
    rays bit-exact against the CPU restatement, rendered image to the parity bar."""
    rho, sp, org = scenes.bos_volume(48)
    g, o = photon.volume_from_density(rho, sp, org, 1), oracle.volume_from_density(rho, sp, org, 1, tex_frac_bits=8)
    g.set_weight_bits(8)
    i = g.info()
    rng = np.random.default_rng(23)
    coords = np.stack([rng.uniform(-1, i.nx + 1, 20000), rng.uniform(-1, i.ny + 1, 20000), rng.uniform(-1, i.nz + 1, 20000)], 1)
    assert_bit_equal(g.sample(coords), o.sample(coords), "8-bit-weight sampler")
    lo, hi = np.array(i.min_bound), np.array(i.max_bound)
    pos = np.stack([rng.uniform(lo[a], hi[a], 3000) for a in range(3)], 1)
    pos[:, 2] = hi[2] + 2000.0
    d = np.stack([rng.normal(0, 0.05, 3000), rng.normal(0, 0.05, 3000), -np.ones(3000)], 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for algorithm in (1, 2):
        gp, gd, gs = g.trace_rays(pos, d, algorithm)
        op, od, os_ = o.trace_rays(pos, d, algorithm)
        assert np.array_equal(gs, os_)
        assert_bit_equal(gp, op, "positions, 8-bit weights")
        assert_bit_equal(gd, od, "directions, 8-bit weights")
    g.set_weight_bits(0)
    exact_p, _, _ = g.trace_rays(pos, d, 2)
    assert not np.array_equal(exact_p, gp)
    g.free(); o.free()
    # through the ABI
    call = scenes.bos_scene(n_dots=8, points_per_dot=20, rays_per_source=100, density_grad_filename=small_volume_file)
    monkeypatch.setenv("PHOTON_INTERP", "linear")
    monkeypatch.setenv("PHOTON_TEX_WEIGHTS", "exact")
    exact = photon.render(call)
    ref_exact, _ = oracle.render(call, interpolation=1, tex_frac_bits=0)
    assert rel_l2(exact, ref_exact) <= IMAGE_TOL, rel_l2(exact, ref_exact)
    monkeypatch.setenv("PHOTON_TEX_WEIGHTS", "fixed8")
    fixed = photon.render(call)
    ref, _ = oracle.render(call, interpolation=1, tex_frac_bits=8)
    assert rel_l2(fixed, ref) <= IMAGE_TOL, rel_l2(fixed, ref)
    assert not np.array_equal(fixed, exact)                     # (a smooth field: the two differ by ~1e-5 rel. L2)
    monkeypatch.delenv("PHOTON_TEX_WEIGHTS")                    # the default is the texture unit's arithmetic
    assert np.array_equal(photon.render(call), fixed)


def test_ray_order_does_not_change_the_image(photon, oracle, small_volume_file, monkeypatch):
    """photon_scene_set_ray_order / PHOTON_RAY_ORDER: source-major (the reference's thread order), lens-major over
    Morton-sorted sources (what keeps waves coherent for full-aperture cones through a volume) and the automatic
    choice must render the same image -- also with position noise, whose generator is keyed by the ray's identity,
    not by its launch slot -- and sharding the sorted order must still cover every source exactly once."""
    import torch
    monkeypatch.setenv("PHOTON_INTERP", "cubic")
    monkeypatch.setenv("PHOTON_NOISE_SEED", "11")
    oracle.set_noise_seed(11)
    call = scenes.piv_scene(n_particles=700, rays_per_source=40, mie=True, polydisperse=True,
                            density_grad_filename=small_volume_file, field_half_width=3.0e4, seed=5)
    call.add_pos_noise, call.pos_noise_std = True, 0.3
    o, st = oracle.render(call, interpolation=2)
    assert st.rk_iterations > 0 and o.any()
    images = {}
    for order in ("source", "lens", "auto"):
        monkeypatch.setenv("PHOTON_RAY_ORDER", order)
        images[order] = photon.render(call)
        assert rel_l2(images[order], o) <= IMAGE_TOL, (order, rel_l2(images[order], o))
    assert rel_l2(images["lens"], images["source"]) <= IMAGE_TOL
    oracle.set_noise_seed(0)
    # device-resident API: lens-major shards [0, n/3) + [n/3, n) of the sorted order == one pass
    call.add_pos_noise = False
    vol = photon.volume_load_nrrd(small_volume_file, 2)
    scene = photon.scene_create(call)
    scene.set_ray_order(1)
    H, W = call.image_shape
    whole = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    parts = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    scene.trace(whole.data_ptr(), vol, 2)
    cut = call.num_sources // 3
    scene.trace(parts.data_ptr(), vol, 2, 0, cut)
    scene.trace(parts.data_ptr(), vol, 2, cut, call.num_sources)
    assert rel_l2(parts.cpu().numpy(), whole.cpu().numpy()) <= IMAGE_TOL
    scene.set_ray_order(0)
    ref = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    scene.trace(ref.data_ptr(), vol, 2)
    assert rel_l2(whole.cpu().numpy(), ref.cpu().numpy()) <= IMAGE_TOL
    scene.free(); vol.free()


@pytest.mark.parametrize("case", ["bos_im1", "bos_full_im1"])
def test_bos_sources_on_device_match_reference_capture(photon, golden_dir, case):
    """photon_sources_bos fed with the reference generator's own dot centres and point template
    (generate_bos_lightfield_data / calculate_sunflower_coordinates, run_simulation_02.py:1328-1551, 999-1056, captured
    by make_golden.py) reproduces, bit for bit, the source arrays the reference's marshalling handed to
    start_ray_tracing for the sample BOS case (shrunk, and at its real size of 120 000 sources)."""
    d = np.load(os.path.join(golden_dir, f"abi_{case}.npz"))
    src = photon.sources_bos(np.stack([d["dot_x"], d["dot_y"]], 1), np.stack([d["tmpl_x"], d["tmpl_y"]], 1),
                             float(d["src_z"][0]), float(d["src_radiance"][0]))
    got = src.download()
    src.free()
    assert got["x"].size == d["src_x"].size
    for key, ref in (("x", "src_x"), ("y", "src_y"), ("z", "src_z"), ("radiance", "src_radiance"),
                     ("diameter_index", "src_diameter_index")):
        assert np.array_equal(got[key], d[ref]), key


def _centroid(img):
    yy, xx = np.mgrid[0:img.shape[0], 0:img.shape[1]]
    w = img.astype(np.float64)
    return np.array([(w * xx).sum(), (w * yy).sum()]) / w.sum()


def test_element_train(photon, oracle, monkeypatch):
    """The working multi-element train (PHOTON_ELEMENT_TRAIN=sequential / photon_scene_set_element_train;
    the reference's device code for it is a stub, .cu:1049-1272, 1331-1333).  Pinned by optics -- two thin
    lenses of focal length 2f in contact image like one of focal length f -- and by GPU-vs-oracle parity,
    including a group of simultaneous elements (two lenslets side by side: rays pick the nearer one)."""
    import copy
    single = scenes.bos_scene(n_dots=5, points_per_dot=20, rays_per_source=100, lens_model="thin-lens", seed=4)
    ref_img = photon.render(single)
    # doublet: same place, half the power each, system indices 2 (hit first) and 1
    doublet = copy.deepcopy(single)
    e = copy.deepcopy(single.elements[0])
    e["element_properties"]["thin_lens_focal_length"] *= 2.0
    doublet.elements = [e, copy.deepcopy(e)]
    doublet.element_center = np.repeat(single.element_center, 2, axis=0)
    doublet.element_plane_parameters = np.repeat(single.element_plane_parameters, 2, axis=0)
    doublet.element_system_index = np.array([2, 1], np.int32)
    monkeypatch.setenv("PHOTON_ELEMENT_TRAIN", "sequential")
    oracle.set_element_train(1)
    try:
        g = photon.render(doublet)
        o, st = oracle.render(doublet)
        assert st.rays_on_sensor > 0
        assert rel_l2(g, o) <= IMAGE_TOL, rel_l2(g, o)
        assert np.abs(_centroid(g) - _centroid(ref_img)).max() < 0.05       # pixels
        assert rel_l2(g, ref_img) < 0.05
        # lenslet pair: two thin lenses side by side on one plane, one system index
        pair = copy.deepcopy(single)
        pitch = single.elements[0]["element_geometry"]["pitch"]
        half = copy.deepcopy(single.elements[0])
        half["element_geometry"]["pitch"] = pitch / 2.0
        pair.elements = [half, copy.deepcopy(half)]
        c = single.element_center[0]
        pair.element_center = np.array([[c[0] - pitch / 4.0, c[1], c[2]], [c[0] + pitch / 4.0, c[1], c[2]]])
        pair.element_plane_parameters = np.repeat(single.element_plane_parameters, 2, axis=0)
        pair.element_system_index = np.array([1, 1], np.int32)
        g2 = photon.render(pair)
        o2, st2 = oracle.render(pair)
        assert 0 < st2.rays_on_sensor < pair.num_rays                          # lenslet apertures clip rays
        assert rel_l2(g2, o2) <= IMAGE_TOL, rel_l2(g2, o2)
        assert rel_l2(g2, ref_img) > 0.1                                       # two displaced images, not one
        # the device-resident API takes the same switch
        import torch
        scene = photon.scene_create(pair)
        scene.set_element_train(1)
        img = torch.zeros(pair.image_shape[0] * pair.image_shape[1], dtype=torch.float32, device="cuda")
        scene.trace(img.data_ptr())
        assert rel_l2(img.cpu().numpy().reshape(pair.image_shape), g2) <= IMAGE_TOL
        scene.free()
    finally:
        oracle.set_element_train(0)
    # default = the reference as it runs: element 0 for every single-member group, nothing for larger groups
    monkeypatch.delenv("PHOTON_ELEMENT_TRAIN")
    g3 = photon.render(doublet)
    o3, _ = oracle.render(doublet)
    assert rel_l2(g3, o3) <= IMAGE_TOL if o3.any() else not g3.any()
    g4 = photon.render(pair)                                                   # one group of two: the stub, rays pass straight
    o4, _ = oracle.render(pair)
    assert rel_l2(g4, o4) <= IMAGE_TOL if o4.any() else not g4.any()


@pytest.mark.parametrize("case", ["a_fwd", "a_swap", "a_rev", "b_stop_lens", "c_three", "d_lens_stop"])
def test_element_train_vs_reference_numpy(photon, oracle, monkeypatch, tmp_path, case):
    """GPU twin of tests/test_oracle_golden.py::test_element_train_vs_reference_numpy: the scene goes through
    start_ray_tracing with PHOTON_ELEMENT_TRAIN=sequential and dumps where each ray meets the sensor (pos_0000.bin,
    .cu:2183-2240); the reference's own numpy sequencer (perform_ray_tracing_03.py:1419-1485) propagated the very same
    rays in float64 (tests/golden/train_f64.npz).  Bars: destroyed-ray masks equal, sensor hits within 0.1 micron (f32
    against f64 over ~1e5 micron of travel; a pixel is 17), and bit-exact against the oracle's dump."""
    from conftest import train_cases
    call = train_cases()[case]
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_f64.npz"))
    call.save_lightrays, call.num_lightrays_save = True, call.num_rays
    monkeypatch.setenv("PHOTON_ELEMENT_TRAIN", "sequential")
    oracle.set_element_train(1)
    dumps = {}
    try:
        for tag, run in (("gpu", lambda c: photon.render(c)), ("cpu", lambda c: oracle.render(c)[0])):
            pdir, ddir = tmp_path / f"{tag}_pos", tmp_path / f"{tag}_dir"
            pdir.mkdir()
            ddir.mkdir()
            call.lightray_position_save_path, call.lightray_direction_save_path = str(pdir), str(ddir)
            img = run(call)
            dumps[tag] = (np.fromfile(pdir / "pos_0000.bin", np.float32).reshape(-1, 3),
                          np.fromfile(ddir / "dir_0000.bin", np.float32).reshape(-1, 3), img)
    finally:
        oracle.set_element_train(0)
    pos, dirs, img = dumps["gpu"]
    assert pos.shape[0] == call.num_rays
    # the dumped direction is the generated one (.cu:2136-2141): the fixture's rays are the rays this render made
    assert_bit_equal(dirs, g[f"{case}_in"][:, 3:6], "generated directions")
    ref_xy = g[f"{case}_sensor_xy"]
    dead_ref = np.isnan(ref_xy[:, 0])
    half = call.camera["pixel_pitch"] * (call.camera["x_pixel_number"] - 1) / 2.0
    assert (np.abs(ref_xy[~dead_ref]) < half - 20.0).all()               # every surviving ray lands well inside the sensor
    assert np.array_equal(np.isnan(pos[:, 0]), dead_ref)
    assert 0.2 < dead_ref.mean() < 0.7
    assert np.abs(pos[~dead_ref, 0:2].astype(np.float64) - ref_xy[~dead_ref]).max() < 0.1
    assert_bit_equal(pos, dumps["cpu"][0], "sensor positions, GPU vs oracle")
    assert img.any() and rel_l2(img, dumps["cpu"][2]) <= IMAGE_TOL


def test_in_library_device_sharding(photon, oracle, small_volume_file, monkeypatch):
    """PHOTON_DEVICES: start_ray_tracing shards the sources over the listed devices (one host thread each,
    private images, one sum at the end -- SURVEY 8e).  One GPU here, so the list repeats device 0: three
    shards side by side must reproduce the single-pass image to the parity bar and keep accumulating on the
    caller's image."""
    call = scenes.bos_scene(n_dots=7, points_per_dot=20, rays_per_source=120, density_grad_filename=small_volume_file)
    monkeypatch.setenv("PHOTON_INTERP", "cubic")
    monkeypatch.delenv("PHOTON_DEVICES", raising=False)
    single = photon.render(call)
    monkeypatch.setenv("PHOTON_DEVICES", "0,0,0")
    base = np.full(call.image_shape, 2.0, np.float32)
    sharded = photon.render(call, base.copy())
    assert rel_l2(sharded - base, single) <= IMAGE_TOL
    o, _ = oracle.render(call, interpolation=2)
    assert rel_l2(sharded - base, o) <= 2 * IMAGE_TOL           # (the +2.0 base costs a little f32 resolution)
    monkeypatch.setenv("PHOTON_DEVICES", "all")                  # whatever the box has
    assert rel_l2(photon.render(call), single) <= IMAGE_TOL
    monkeypatch.setenv("PHOTON_DEVICES", "0,99")                 # a device that is not there: falls back, says so
    assert rel_l2(photon.render(call), single) <= IMAGE_TOL
    # each device uploads only its block of the sources; the noise generator is keyed by the ray's place in the
    # CALLER's source list, so the shards draw the same numbers as the single pass
    call.add_pos_noise, call.pos_noise_std = True, 0.4
    monkeypatch.setenv("PHOTON_NOISE_SEED", "7")
    monkeypatch.delenv("PHOTON_DEVICES")
    noisy = photon.render(call)
    assert rel_l2(noisy, single) > 1e-3                          # the noise does something
    monkeypatch.setenv("PHOTON_DEVICES", "0,0,0")
    assert rel_l2(photon.render(call), noisy) <= IMAGE_TOL


def test_doomed_rays_are_skipped_not_missed(photon, oracle, small_volume_file, monkeypatch):
    """photon_scene_set_skip_doomed / PHOTON_SKIP_DOOMED: rays aimed outside the first aperture by more than the
    volume can bend them back are dropped before the march.  Image and rays_on_sensor must not change (thick and
    thin lens, both ray orders); the march does visibly less work for a full-aperture cone and exactly the same
    for a BOS cone (nothing doomed)."""
    import torch
    monkeypatch.setenv("PHOTON_INTERP", "cubic")
    vol = photon.volume_load_nrrd(small_volume_file, 2)
    for lens_model in ("general", "thin-lens"):
        call = scenes.piv_scene(n_particles=500, rays_per_source=64, mie=False, density_grad_filename=small_volume_file,
                                field_half_width=2.5e4, seed=9)
        if lens_model == "thin-lens":
            geom = scenes.single_lens_camera(lens_model="thin-lens", **scenes.SAMPLE_LENS)
            call.elements = [geom["element"]]
        o, ost = oracle.render(call, interpolation=2)
        H, W = call.image_shape
        scene = photon.scene_create(call)
        res = {}
        for skip in (0, 1):
            for order in (0, 1):
                scene.set_skip_doomed(skip)
                scene.set_ray_order(order)
                img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
                st = scene.trace(img.data_ptr(), vol, 2, want_stats=True)
                res[(skip, order)] = (img.cpu().numpy().reshape(H, W), st)
                assert rel_l2(res[(skip, order)][0], o) <= IMAGE_TOL, (lens_model, skip, order)
                assert st.rays_on_sensor == ost.rays_on_sensor
        assert res[(0, 0)][1].rk_iterations == ost.rk_iterations
        assert res[(1, 0)][1].rk_iterations < 0.9 * ost.rk_iterations           # a good part of the cone is doomed
        assert res[(0, 0)][1].rays_marched == call.num_rays                     # stats say how many rays were really marched
        assert 0 < res[(1, 0)][1].rays_marched < 0.9 * call.num_rays and res[(1, 1)][1].rays_marched == res[(1, 0)][1].rays_marched
        assert res[(1, 1)][1].rk_iterations == res[(1, 0)][1].rk_iterations
        scene.free()
    bos = scenes.bos_scene(n_dots=5, points_per_dot=20, rays_per_source=100, density_grad_filename=small_volume_file)
    scene = photon.scene_create(bos)
    img = torch.zeros(bos.image_shape[0] * bos.image_shape[1], dtype=torch.float32, device="cuda")
    _, ost = oracle.render(bos, interpolation=2)
    assert scene.trace(img.data_ptr(), vol, 2, want_stats=True).rk_iterations == ost.rk_iterations
    scene.free(); vol.free()
    monkeypatch.setenv("PHOTON_SKIP_DOOMED", "0")                         # the ABI path takes the switch from the environment
    assert rel_l2(photon.render(call), o) <= IMAGE_TOL


@pytest.mark.parametrize("seed", range(8))
def test_randomised_scenes(photon, oracle, small_volume_file, monkeypatch, seed):
    """A sweep over the knobs that decide which code path a wave takes -- rays per source around the wave size,
    cone width from a micron to the full aperture (coherent tiles, bricks, gather), sampler, integrator, ray
    order, BOS vs PIV sensor -- each case against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    rays = int(rng.choice([1, 2, 31, 63, 64, 65, 100, 257]))
    ratio = float(rng.choice([1e-4, 1e-2, 0.1, 1.0]))
    interp = int(rng.choice([1, 2]))
    algorithm = int(rng.choice([1, 2]))
    order = str(rng.choice(["source", "lens", "auto"]))
    if rng.random() < 0.5:
        call = scenes.bos_scene(n_dots=int(rng.integers(3, 12)), points_per_dot=int(rng.integers(5, 30)), rays_per_source=rays,
                                density_grad_filename=small_volume_file, ray_tracing_algorithm=algorithm, seed=seed)
    else:
        call = scenes.piv_scene(n_particles=int(rng.integers(50, 600)), rays_per_source=rays, mie=bool(rng.integers(0, 2)),
                                polydisperse=True, density_grad_filename=small_volume_file, field_half_width=2.5e4,
                                ray_tracing_algorithm=algorithm, seed=seed)
    call.ray_cone_pitch_ratio = ratio
    monkeypatch.setenv("PHOTON_RAY_ORDER", order)
    g, o, st = _render_both(photon, oracle, call, interp, monkeypatch)
    assert st.rk_iterations > 0
    if o.any():
        assert rel_l2(g, o) <= IMAGE_TOL, (seed, rays, ratio, interp, algorithm, order, rel_l2(g, o))
    else:
        assert not g.any()


def test_api_misuse_is_reported_not_fatal(photon, capfd):
    """Bad arguments to the extension entry points come back as non-zero return codes (raised as PhotonError by
    the ctypes veneer), never as a crash, and leave no half-built handle behind."""
    import ctypes
    from photon_amd.library import PhotonError
    L = photon.lib
    h = ctypes.c_void_p()
    assert L.photon_volume_load_nrrd(b"/nonexistent.nrrd", 1, ctypes.byref(h)) != 0 and not h.value
    rho = np.ones((2, 2, 2), np.float32)                                    # fewer than 3 points per axis
    with pytest.raises(PhotonError):
        photon.volume_from_density(rho, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), 1)
    with pytest.raises(PhotonError):
        photon.volume_from_density(np.ones((4, 4, 4), np.float32), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), 3)   # unknown sampler
    with pytest.raises(PhotonError):
        photon.volume_gaussian(8, 1.0, (0, 0, 0), 1.0, 0.1, (0, 0, 0), 0.0, 1)                              # sigma = 0
    with pytest.raises(PhotonError):
        photon.sources_piv(1, -5, (0, 0, 0), (1, 1, 1), 0.0, 730.0, 1.0)                                    # negative count
    with pytest.raises(PhotonError):
        photon.sources_piv(1, 5, (0, 0, 0), (1, 1, 1), 0.0, 0.0, 1.0)                                       # beam width 0
    with pytest.raises(PhotonError):
        photon.sources_bos(np.zeros((2, 2)), np.zeros((0, 2)), 0.0, 1.0)                                    # empty template
    assert L.photon_sources_count(None) == -1
    vol = photon.volume_from_density(np.ones((4, 4, 4), np.float32), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), 1)
    assert L.photon_volume_set_weight_bits(vol.handle, 40) != 0 and L.photon_volume_set_weight_bits(None, 8) != 0
    call = scenes.bos_scene(n_dots=1, points_per_dot=2, rays_per_source=4)
    scene = photon.scene_create(call)
    assert L.photon_scene_set_ray_order(scene.handle, 7) != 0 and L.photon_scene_set_element_train(scene.handle, 2) != 0
    img = np.zeros(call.image_shape, np.float32)
    assert L.photon_trace(scene.handle, None, 0, 0, call.num_sources + 1, img.ctypes.data_as(ctypes.c_void_p), None, None) != 0
    assert L.photon_trace(None, None, 0, 0, 0, None, None, None) != 0
    scene.free(); vol.free()
    assert "photon:" in capfd.readouterr().err


def test_errors_leave_image_untouched(photon, small_volume_file, capfd):
    call = scenes.bos_scene(n_dots=2, points_per_dot=5, rays_per_source=8, density_grad_filename=small_volume_file)
    img = np.full(call.image_shape, 1.5, np.float32)
    call.density_grad_filename = "/nonexistent/volume.nrrd"
    assert np.array_equal(photon.render(call, img.copy()), img)
    assert "photon:" in capfd.readouterr().err


def test_c2_full_size_against_oracle(photon, oracle):
    """BASELINE config C2 at its full size (1e6 rays, thick lens, Mie, 4-pixel splat): the oracle
    finishes this one in seconds, so it is compared directly."""
    call = scenes.config("C2")
    assert call.num_rays == 1_000_000
    g, o, st = _render_both(photon, oracle, call)
    assert st.rays_on_sensor > 100_000
    assert rel_l2(g, o) <= IMAGE_TOL, rel_l2(g, o)


# ------------------------------------------------------------------------------------------------
# BASELINE-size properties (no oracle: size-independent invariants)
# ------------------------------------------------------------------------------------------------
def test_full_size_c3_properties(photon, oracle, workdir, monkeypatch):
    """1e7 rays through the 256^3 volume (tricubic RK4): (1) sharding the sources in two and
    summing reproduces the single-pass image (what the multi-GPU path relies on); (2) a uniform
    volume deflects nothing: the image equals the no-volume image; (3) counters add up; (4) a 40-source slice of the
    job equals the oracle's render of those sources through the same 256^3 volume, for both samplers."""
    import torch
    monkeypatch.setenv("PHOTON_INTERP", "cubic")
    call = scenes.config("C3", workdir)
    assert call.num_rays == 10_000_000
    scene = photon.scene_create(call)
    vol = photon.volume_load_nrrd(call.density_grad_filename, 2)
    H, W = call.image_shape
    full = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    st = scene.trace(full.data_ptr(), vol, 2, want_stats=True)
    assert st.rays_launched == call.num_rays
    assert st.rk_iterations >= 250 * call.num_rays and st.volume_samples >= 3 * st.rk_iterations
    assert st.rays_on_sensor == call.num_rays and st.sensor_taps > 10 * call.num_rays
    # a 40-source slice of the same 256^3 job against the oracle (~1 s of oracle each): tricubic, and trilinear with the
    # 8-bit weights -- the sampler the reference executes (interpolation_scheme hard-coded 1, parallel_ray_tracing.cu:3330;
    # trace_rays_through_density_gradients.h:992-1181)
    head = scenes.config("C3", workdir)
    for f in ("src_x", "src_y", "src_z", "src_radiance", "src_diameter_index"):
        setattr(head, f, getattr(head, f)[:40])
    for interp in (2, 1):
        v = vol if interp == 2 else photon.volume_load_nrrd(call.density_grad_filename, 1)
        sl = torch.zeros(H * W, dtype=torch.float32, device="cuda")
        sst = scene.trace(sl.data_ptr(), v, 2, 0, 40, want_stats=True)
        torch.cuda.synchronize()
        ref, ost = oracle.render(head, interpolation=interp)
        assert ost.rk_iterations >= 250 * head.num_rays and sst.rk_iterations == ost.rk_iterations
        assert ref.any() and rel_l2(sl.cpu().numpy().reshape(H, W), ref) <= IMAGE_TOL, interp
        if interp == 1:
            v.free()
    halves = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    mid = call.num_sources // 2 + 7
    scene.trace(halves.data_ptr(), vol, 2, 0, mid)
    scene.trace(halves.data_ptr(), vol, 2, mid, call.num_sources)
    torch.cuda.synchronize()
    assert rel_l2(halves.cpu().numpy(), full.cpu().numpy()) <= IMAGE_TOL
    # uniform density -> zero gradient -> straight rays
    rho = np.full((32, 32, 32), 1.225, np.float32)
    _, sp, org = scenes.bos_volume(32)
    uni = photon.volume_from_density(rho, sp, org, 2)
    a = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    b = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    scene.trace(a.data_ptr(), uni, 2, 0, 2000)
    scene.trace(b.data_ptr(), None, 0, 0, 2000)
    torch.cuda.synchronize()
    assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 1e-3      # f32 world-transform round trip only
    scene.free()
    vol.free()
    uni.free()


def test_c4_one_gpu_share_properties(photon, workdir, monkeypatch):
    """BASELINE config C4 (1e8 rays, 512^3, 8 GPUs): one GPU's share -- 1.25e7 rays through the full
    512^3 volume (2 GiB of texels + 2 GiB of B-spline coefficients in HBM).  Checked through
    invariants: every ray crosses the whole grid (>= 509 RK4 iterations), splitting the sources
    reproduces the single-pass image, and the trilinear and tricubic renders of this smooth field
    agree to a fraction of a percent."""
    import torch
    call = scenes.config("C4", workdir, scale=0.125, volume_n=512)
    assert call.num_rays == 12_500_000
    H, W = call.image_shape
    scene = photon.scene_create(call)
    images = {}
    for interp in (2, 1):
        vol = photon.volume_load_nrrd(call.density_grad_filename, interp)
        i = vol.info()
        assert (i.nx, i.ny, i.nz) == (512, 512, 512)
        img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
        st = scene.trace(img.data_ptr(), vol, 2, want_stats=True)
        assert st.rk_iterations >= 509 * call.num_rays and st.rays_on_sensor == call.num_rays
        if interp == 2:
            parts = torch.zeros(H * W, dtype=torch.float32, device="cuda")
            cut = call.num_sources // 3
            scene.trace(parts.data_ptr(), vol, 2, 0, cut)
            scene.trace(parts.data_ptr(), vol, 2, cut, call.num_sources)
            torch.cuda.synchronize()
            assert rel_l2(parts.cpu().numpy(), img.cpu().numpy()) <= IMAGE_TOL
        images[interp] = img.cpu().numpy()
        vol.free()
    assert 0 < rel_l2(images[1], images[2]) < 2e-2
    scene.free()


def test_c4_whole_job_on_one_gpu(photon, oracle, workdir, monkeypatch):
    """BASELINE config C4 WHOLE: 2e5 sources x 500 = 1e8 rays through the 512^3 volume (tricubic RK4) on one GPU -- the
    multi-launch loop WITH a volume (two launches of <= 2^26 rays; the reference chunks at 10^4 sources,
    parallel_ray_tracing.cu:3366-3372, 3515-3558): (1) device-resident photon_trace: every ray crosses the grid (>= 509
    iterations) and lands; (2) the same job through start_ray_tracing (host arrays in, host image out) gives the same
    image; (3) so do eight shards side by side (PHOTON_DEVICES=0 x 8: the per-device upload / accumulate / sum path);
    (4) a 20-source slice equals the oracle's render through the same 512^3 volume -- tricubic, and trilinear with 8-bit
    weights (what the reference executes)."""
    import torch
    monkeypatch.setenv("PHOTON_INTERP", "cubic")
    call = scenes.config("C4", workdir, volume_n=512)
    assert call.num_rays == 100_000_000 and call.num_sources == 200_000
    H, W = call.image_shape
    scene = photon.scene_create(call)
    vol = photon.volume_load_nrrd(call.density_grad_filename, 2)
    img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    st = scene.trace(img.data_ptr(), vol, 2, want_stats=True)
    assert st.rays_launched == call.num_rays and st.rays_on_sensor == call.num_rays and st.rays_marched == call.num_rays
    assert st.rk_iterations >= 509 * call.num_rays
    whole = img.cpu().numpy().reshape(H, W)
    sl = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    scene.trace(sl.data_ptr(), vol, 2, 0, 20)
    torch.cuda.synchronize()
    slice_gpu = sl.cpu().numpy().reshape(H, W)
    vol.free()
    lin = photon.volume_load_nrrd(call.density_grad_filename, 1)         # the reference's executed sampler, same slice
    sl.zero_()
    scene.trace(sl.data_ptr(), lin, 2, 0, 20)
    torch.cuda.synchronize()
    slice_gpu_linear = sl.cpu().numpy().reshape(H, W)
    lin.free()
    scene.free()
    del img, sl
    torch.cuda.empty_cache()
    abi = photon.render(call)                                          # the reference's entry point, two launches
    assert rel_l2(abi, whole) <= IMAGE_TOL
    monkeypatch.setenv("PHOTON_DEVICES", "0,0,0,0,0,0,0,0")
    shards = photon.render(call)
    monkeypatch.delenv("PHOTON_DEVICES")
    assert rel_l2(shards, whole) <= IMAGE_TOL
    head = scenes.config("C4", workdir, volume_n=512)
    for f in ("src_x", "src_y", "src_z", "src_radiance", "src_diameter_index"):
        setattr(head, f, getattr(head, f)[:20])
    ref, ost = oracle.render(head, interpolation=2)
    assert ost.rk_iterations >= 509 * head.num_rays
    assert rel_l2(slice_gpu, ref) <= IMAGE_TOL
    ref, ost = oracle.render(head, interpolation=1)                      # trilinear, 8-bit weights (.cu:3330, .h:992-1181)
    assert ost.rk_iterations >= 509 * head.num_rays
    assert ref.any() and rel_l2(slice_gpu_linear, ref) <= IMAGE_TOL


@pytest.mark.parametrize("case", ["piv", "bos_im1", "bos_im2"])
def test_postprocess_on_device_matches_reference(photon, golden_dir, case):
    """photon_postprocess_u16 (SURVEY 8f rank 1 behind the C-ABI): the uint16 picture the reference's own
    post-processing (perform_ray_tracing_03.py:2190-2247) made of a synthetic raw image, reproduced on the device pixel
    for pixel; then the centre crop (:2250-2259) and the no-rescaling branch against the numpy mirror."""
    import json
    import torch
    from photon_amd.ray_tracing import crop_window, postprocess_image
    a = np.load(os.path.join(golden_dir, f"postprocess_{case}.npz"))
    with open(os.path.join(golden_dir, f"abi_{case}.json")) as f:
        pp = json.load(f)["postprocess"]
    H, W = (int(v) for v in a["shape"])
    raw = np.zeros(H * W, np.float32)
    raw[a["raw_index"]] = a["raw_value"]
    raw[11] = np.nan                                                    # non-finite pixels are zeroed (:2228)
    ref = np.zeros(H * W, np.uint16)
    ref[a["out_u16_index"]] = a["out_u16_value"]
    d_raw = torch.from_numpy(raw).cuda()
    d_out = torch.zeros(H * W, dtype=torch.int16, device="cuda")
    rows, cols = photon.postprocess_u16(d_raw.data_ptr(), W, H, d_out.data_ptr(), pp["pixel_gain"], int(pp["pixel_bit_depth"]),
                                        pp["intensity_rescaling"])
    assert (rows, cols) == (H, W)
    got = d_out.cpu().numpy().view(np.uint16)
    assert np.array_equal(got, ref)
    assert np.array_equal(d_raw.cpu().numpy(), raw, equal_nan=True)     # without noise the raw image is only read
    # crop: the reference's window, one row / column short of what was asked
    rows, cols = photon.postprocess_u16(d_raw.data_ptr(), W, H, d_out.data_ptr(), pp["pixel_gain"], int(pp["pixel_bit_depth"]),
                                        True, crop_rows=300, crop_cols=200)
    rs, cs = crop_window(H, W, 300, 200)
    assert (rows, cols) == (299, 199) == (rs.stop - rs.start, cs.stop - cs.start)
    want = postprocess_image(np.nan_to_num(raw, nan=0.0).reshape(H, W), pp["pixel_gain"], int(pp["pixel_bit_depth"]))[rs, cs]
    assert np.array_equal(d_out.cpu().numpy().view(np.uint16)[:rows * cols].reshape(rows, cols), want)
    # intensity_rescaling off: clip and convert only
    photon.postprocess_u16(d_raw.data_ptr(), W, H, d_out.data_ptr(), pp["pixel_gain"], int(pp["pixel_bit_depth"]), False)
    want = postprocess_image(np.nan_to_num(raw, nan=0.0).reshape(H, W), pp["pixel_gain"], int(pp["pixel_bit_depth"]), False)
    assert np.array_equal(d_out.cpu().numpy().view(np.uint16).reshape(H, W), want)
    # seeded sensor noise: rewrites the raw image like the reference does, reproducibly
    photon.postprocess_u16(d_raw.data_ptr(), W, H, d_out.data_ptr(), pp["pixel_gain"], int(pp["pixel_bit_depth"]), True,
                           image_noise=0.05, noise_seed=3)
    noisy = d_raw.cpu().numpy()
    delta = (noisy - raw)[np.isfinite(raw)]
    assert abs(float(delta.std()) - 5.0) < 0.05 and abs(float(delta.mean())) < 0.05
    with pytest.raises(Exception):
        photon.postprocess_u16(d_raw.data_ptr(), W, H, d_out.data_ptr(), 0.0, 10, True, crop_rows=4 * H, crop_cols=10)


def test_c5_full_size_properties(photon, oracle, workdir, monkeypatch):
    """BASELINE config C5 at its full size on one GPU: 1e6 polydisperse Mie particles x 40 rays = 4e7 rays through the
    256^3 volume (tricubic RK4, full-aperture cones -> lens-major order over device-sorted sources, doomed rays skipped).
    Invariants at full size -- two shards = whole (what the 8-GPU split relies on), doomed-ray skip and ray order leave
    image and rays_on_sensor unchanged on a 1e5-particle cut, the statistics add up -- and a leading slice of the same
    particle list against the oracle."""
    import torch
    monkeypatch.setenv("PHOTON_INTERP", "cubic")
    call = scenes.config("C5", workdir)
    assert call.num_sources == 1_000_000 and call.num_rays == 40_000_000
    scene = photon.scene_create(call)
    vol = photon.volume_load_nrrd(call.density_grad_filename, 2)
    H, W = call.image_shape
    full = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    st = scene.trace(full.data_ptr(), vol, 2, want_stats=True)
    assert st.rays_launched == call.num_rays
    assert 0.5 * call.num_rays < st.rays_marched < call.num_rays          # part of every full-aperture cone is doomed
    assert 0.3 * call.num_rays < st.rays_on_sensor < st.rays_marched
    assert st.rk_iterations > 150 * st.rays_on_sensor
    parts = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    cut = call.num_sources // 2 + 13
    scene.trace(parts.data_ptr(), vol, 2, 0, cut)
    scene.trace(parts.data_ptr(), vol, 2, cut, call.num_sources)
    torch.cuda.synchronize()
    assert rel_l2(parts.cpu().numpy(), full.cpu().numpy()) <= IMAGE_TOL
    # ray order / doomed-ray skip on the first 1e5 particles
    ref = None
    for order, skip in ((1, 1), (0, 1), (1, 0)):
        scene.set_ray_order(order)
        scene.set_skip_doomed(skip)
        img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
        s2 = scene.trace(img.data_ptr(), vol, 2, 0, 100_000, want_stats=True)
        if ref is None:
            ref = (img.cpu().numpy(), s2.rays_on_sensor)
        else:
            assert rel_l2(img.cpu().numpy(), ref[0]) <= IMAGE_TOL and s2.rays_on_sensor == ref[1], (order, skip)
    scene.free()
    vol.free()
    # the leading 1500 particles through the C-ABI against the oracle
    for f in ("src_x", "src_y", "src_z", "src_radiance", "src_diameter_index"):
        setattr(call, f, getattr(call, f)[:1500])
    g = photon.render(call)
    o, ost = oracle.render(call, interpolation=2)
    assert ost.rays_on_sensor > 10_000 and rel_l2(g, o) <= IMAGE_TOL, rel_l2(g, o)


def test_nrrd_header_cannot_drive_the_allocation(photon, tmp_path):
    """A header whose sizes promise more data than the file holds (or absurd sizes) is rejected with an error code
    before anything is allocated -- no exception crosses the C boundary."""
    import ctypes
    L = photon.lib
    for sizes in ("60000 60000 60000", "70000 4 4", "64 64 64"):
        p = tmp_path / "bad.nrrd"
        p.write_bytes((f"NRRD0005\ntype: float\ndimension: 3\nsizes: {sizes}\nendian: little\nencoding: raw\n"
                       "spacings: 1 1 1\nspace origin: (0,0,0)\n\n").encode() + b"\0" * 4096)
        h = ctypes.c_void_p()
        assert L.photon_volume_load_nrrd(str(p).encode(), 1, ctypes.byref(h)) != 0 and not h.value


def test_nrrd_payload_encodings(photon, oracle, tmp_path):
    """The volume file as other tools write it: teem's nrrdLoad -- what the reference reads it with (.h:1687) -- takes gzip
    (pynrrd's default), ascii and either byte order beside photon's own raw little-endian.  Every encoding of one density
    field loads to the SAME texels, bit for bit, as the raw file (and as the oracle's build of the array); damaged or
    dishonest compressed payloads are refused with an error code, nothing allocated on their say-so."""
    import ctypes
    import gzip
    import zlib
    rng = np.random.default_rng(3)
    n = (12, 10, 14)                                                     # nz, ny, nx
    rho = (1.2 + 0.1 * rng.random(n)).astype(np.float32)
    sp, org = (100.0, 110.0, 120.0), (-500.0, -400.0, 300000.0)

    def header(encoding, endian):
        return (f"NRRD0005\ntype: float\ndimension: 3\nsizes: {n[2]} {n[1]} {n[0]}\nendian: {endian}\nencoding: {encoding}\n"
                f"spacings: {sp[0]} {sp[1]} {sp[2]}\nspace origin: ({org[0]},{org[1]},{org[2]})\n\n").encode()
    little, big = rho.astype("<f4").tobytes(), rho.astype(">f4").tobytes()
    files = {"raw_little": header("raw", "little") + little, "raw_big": header("raw", "big") + big,
             "gzip_little": header("gzip", "little") + gzip.compress(little), "gz_big": header("gz", "big") + gzip.compress(big),
             "zlib_wrapped": header("gzip", "little") + zlib.compress(little),
             "ascii": header("ascii", "little") + " ".join(repr(float(v)) for v in rho.ravel()).encode() + b"\n",
             "text_lines": header("text", "little") + "\n".join(repr(float(v)) for v in rho.ravel()).encode()}
    want = None
    for name, blob in files.items():
        path = tmp_path / f"{name}.nrrd"
        path.write_bytes(blob)
        v = photon.volume_load_nrrd(str(path), 1)
        tex = v.download()
        v.free()
        if want is None:
            want = tex
            o = oracle.volume_from_density(rho, sp, org, 1)
            assert_bit_equal(tex, o.download(), "raw file vs the oracle's build of the same array")
            o.free()
        assert_bit_equal(tex, want, name)
    L = photon.lib
    gz = gzip.compress(little)
    bad = {"truncated": header("gzip", "little") + gz[:len(gz) // 2], "garbage": header("gzip", "little") + b"\x1f\x8b" + b"\0" * 64,
           "short": header("gzip", "little") + gzip.compress(little[:-400]),
           "sizes_lie": header("gzip", "little").replace(f"sizes: {n[2]} {n[1]} {n[0]}".encode(), b"sizes: 60000 60000 60000") + gz,
           "ascii_short": header("ascii", "little") + b"1.0 2.0 3.0\n", "unknown": header("bzip2", "little") + little,
           "endian": header("raw", "middle") + little}
    for name, blob in bad.items():
        path = tmp_path / f"bad_{name}.nrrd"
        path.write_bytes(blob)
        h = ctypes.c_void_p()
        assert L.photon_volume_load_nrrd(str(path).encode(), 1, ctypes.byref(h)) != 0 and not h.value, name


def test_normal_range_division_and_sqrt_are_exact(photon):
    """The march loops divide and take square roots with the compiler's correctly rounded sequences MINUS their range scaling
    (device_vec.hpp, div_nr / rcp_nr / sqrt_nr: 16 instructions fewer per RK4 iteration).  Same instructions on the same
    values whenever no scaling would have happened -- so the IEEE result, bit for bit, for every operand and result within
    [2^-96, 2^96]: held here against numpy's float32 division and square root (correctly rounded by IEEE 754) on 3e6 operand
    pairs spread over that whole range, on the march's own neighbourhood (step / n, 1 / n, 1 / |T|), and on the specials
    (zeros, infinities, NaN), which v_div_fixup still handles."""
    import ctypes
    f = photon.lib.photon_selftest_normal_range_math
    f.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5
    rng = np.random.default_rng(21)

    def run(a, b):
        a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
        q, r, s = (np.empty_like(a) for _ in range(3))
        assert f(a.size, *(x.ctypes.data for x in (a, b, q, r, s))) == 0
        return q, r, s

    def bits(x):
        return np.ascontiguousarray(x, np.float32).view(np.uint32)

    # (1) the whole guaranteed range: magnitudes 2^-45 .. 2^45 each, so that quotients stay within 2^-90 .. 2^90
    n = 1_000_000
    a = (rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-45, 46, n) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    b = (rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-45, 46, n) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    q, r, s = run(a, b)
    with np.errstate(invalid="ignore"):
        assert np.array_equal(bits(q), bits(a / b)) and np.array_equal(bits(r), bits(np.float32(1.0) / b))
        pos = a > 0
        assert np.array_equal(bits(s[pos]), bits(np.sqrt(a[pos]))) and np.isnan(s[~pos]).all()
    # (2) square roots over the full guaranteed range 2^-90 .. 2^90
    a = (rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-90, 91, n)).astype(np.float32)
    _, _, s = run(a, np.ones(n, np.float32))
    assert np.array_equal(bits(s), bits(np.sqrt(a)))
    # (3) where a marching ray lives: step / n and 1 / n with n = 1 + (n - 1), |T|^2 around n^2
    nn = (1.0 + rng.uniform(-0.3, 0.3, n)).astype(np.float32)
    step = rng.uniform(50.0, 500.0, n).astype(np.float32)
    q, r, _ = run(step, nn)
    assert np.array_equal(bits(q), bits(step / nn)) and np.array_equal(bits(r), bits(np.float32(1.0) / nn))
    d2 = (nn * nn * rng.uniform(0.98, 1.02, n)).astype(np.float32)
    _, _, s = run(d2, nn)
    assert np.array_equal(bits(s), bits(np.sqrt(d2)))
    # (4) specials: what v_div_fixup returns for them is what IEEE division returns
    a = np.array([0.0, 1.0, -1.0, np.inf, 1.0, 0.0, np.nan, 1.0, np.inf, -0.0], np.float32)
    b = np.array([1.0, 0.0, 0.0, 2.0, np.inf, 0.0, 1.0, np.nan, np.inf, 3.0], np.float32)
    q, r, s = run(a, b)
    with np.errstate(divide="ignore", invalid="ignore"):
        want_q, want_r = a / b, np.float32(1.0) / b
    for got, want in ((q, want_q), (r, want_r)):
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(bits(got[~np.isnan(want)]), bits(want[~np.isnan(want)]))
    assert s[0] == 0.0 and s[1] == 1.0 and np.isnan(s[2]) and np.isinf(s[3])


def _render_skip_pair(photon, call, monkeypatch):
    out = {}
    for skip in ("0", "1"):
        monkeypatch.setenv("PHOTON_SKIP_DOOMED", skip)
        out[skip] = photon.render(call).astype(np.float64)
    monkeypatch.delenv("PHOTON_SKIP_DOOMED")
    return out["0"], out["1"]


@pytest.mark.parametrize("variant", ["sample", "wide_field", "thin_lens", "ratio_0.7", "deep_sheet", "tilted_lens", "flipped_normal"])
def test_dead_lens_samples_are_not_launched_and_nothing_changes(photon, oracle, monkeypatch, variant):
    """The volume-free path launches only the lens samples that can reach the first element's aperture from SOME source
    (photon_scene.hip, live_lens_samples: a bound over all sources of the scene; the reference aims ray k of every source at
    the same point of the lens plane, parallel_ray_tracing.cu:123-141, and kills what meets the front surface beyond pitch / 2,
    :447, 560-566).  A ray that is not launched adds nothing; so the image with the skip must equal the image without it BIT
    FOR BIT (the surviving rays' increments are the same f32 numbers, added in f64) -- with sources far off the axis, a sheet
    as deep as the working distance allows, a thin lens, a cone that only just overfills the aperture -- and must equal the
    oracle's, which launches everything.  A tilted first element is outside what the bound covers: every sample launched."""
    kw = dict(n_particles=90, rays_per_source=3000, mie=True, seed=11)
    if variant == "ratio_0.7":
        kw["ray_cone_pitch_ratio"] = 0.7
    if variant == "wide_field":
        kw["field_half_width"] = 3.0e5                      # sources up to 4.2e5 um off the axis: most miss the sensor, the bound must still hold
    call = scenes.piv_scene(**kw)
    if variant == "thin_lens":
        call.elements[0]["element_type"] = "t"               # its focal length is in the element's properties already
    if variant == "deep_sheet":
        rng = np.random.default_rng(5)
        call.src_z = (call.src_z + rng.uniform(-2.0e5, 2.0e5, call.src_z.size)).astype(call.src_z.dtype)
    if variant == "tilted_lens":
        call.element_plane_parameters = np.array([[0.02, 0.0, 1.0, call.element_plane_parameters[0][3]]])
    if variant == "flipped_normal":                           # plane normal (0, 0, -1): the front sphere moves (.cu:557); the bound does not cover it
        call.element_plane_parameters = -np.asarray(call.element_plane_parameters)
    scene = photon.scene_create(call)
    live = scene.live_rays()
    scene.free()
    rps = call.lightray_number_per_particle
    if variant in ("tilted_lens", "flipped_normal"):
        assert live == rps
    elif variant == "ratio_0.7":
        assert 0.45 * rps < live < 0.85 * rps, live         # radius uniform in [0, 0.7 pitch]: 5/7 inside pitch / 2, plus the margin
    else:
        assert 0.45 * rps < live < 0.62 * rps, live         # radius uniform in [0, pitch]: half inside, plus the margin
    without, with_skip = _render_skip_pair(photon, call, monkeypatch)
    assert without.sum() > 0 or variant in ("wide_field", "flipped_normal")
    assert np.array_equal(with_skip, without), float(np.abs(with_skip - without).max())
    ref, st = oracle.render(call)
    if ref.sum() > 0:
        assert rel_l2(with_skip, ref) <= IMAGE_TOL, rel_l2(with_skip, ref)


@pytest.mark.parametrize("variant", ["sample", "off_centre", "deep_sheet", "chunks", "devices", "thin_lens"])
def test_sources_that_miss_the_sensor_are_not_launched_and_nothing_changes(photon, oracle, monkeypatch, variant):
    """The volume-free path also leaves out the SOURCES whose image cannot fall on the sensor (photon_scene.hip,
    source_misses_sensor: an interval bound on where one biconvex thick lens -- or one thin lens -- can put a source's rays; photon's sample PIV
    frame draws particles over a field 1.5 x wider than the camera sees, run_simulation_02.py:956-958).  The list the scene
    launches is the complement of the host bound (tests/test_source_cull.py holds that against exact ray tracing); the image
    with the skip equals the image without it bit for bit and equals the oracle's, which launches everything -- in one launch,
    in several (photon_trace on sub-ranges: each launch takes its slice of the list; the 5e8-ray sample frame of
    tests/test_sample_full_gpu.py is the multi-launch case of start_ray_tracing) and sharded over devices
    (each shard's scene holds and culls only its own sources)."""
    from exact_lens import lens_samples
    call = scenes.piv_scene(n_particles=1200, rays_per_source=600, mie=True, seed=21)
    rng = np.random.default_rng(8)
    if variant == "off_centre":
        call.src_x = (call.src_x + 6.0e4).astype(call.src_x.dtype)
    if variant == "deep_sheet":
        call.src_z = (call.src_z + rng.uniform(-1.5e5, 1.5e5, call.src_z.size)).astype(call.src_z.dtype)
    if variant == "devices":
        monkeypatch.setenv("PHOTON_DEVICES", "0,0,0")
    if variant == "thin_lens":
        call.elements[0]["element_type"] = "t"                          # photon's thin-lens model: the bound is the exact linear map
    scene = photon.scene_create(call)
    kept = scene.live_sources()
    scene.free()
    off = photon.sources_missing_sensor(call, *lens_samples(photon, call))
    assert kept is not None and np.array_equal(kept, np.flatnonzero(~off))
    assert 0.2 * call.num_sources < kept.size < (0.8 if variant == "off_centre" else 0.6) * call.num_sources, kept.size
    without, with_skip = _render_skip_pair(photon, call, monkeypatch)
    assert without.sum() > 0
    assert np.array_equal(with_skip, without), float(np.abs(with_skip - without).max())
    ref, st = oracle.render(call)
    assert st.rays_on_sensor > 0 and rel_l2(with_skip, ref) <= IMAGE_TOL, rel_l2(with_skip, ref)
    if variant == "chunks":                                             # photon_trace on sub-ranges: each launch takes its slice of the list
        import torch
        scene = photon.scene_create(call)
        H, W = call.image_shape
        img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
        total = 0
        for b, e in ((0, 250), (250, 251), (251, 777), (777, 777), (777, call.num_sources)):
            st = scene.trace(img.data_ptr(), None, 0, b, e, want_stats=True)
            total += st.rays_on_sensor
        torch.cuda.synchronize()
        scene.free()
        assert total == oracle.render(call)[1].rays_on_sensor
        assert rel_l2(img.cpu().numpy().reshape(H, W), without) <= 1e-7
    # what disables it: sensor-position noise (unbounded jitter) and the ray dumps (every ray has a slot)
    noisy = scenes.piv_scene(n_particles=1200, rays_per_source=60, mie=False, seed=21)
    noisy.add_pos_noise, noisy.pos_noise_std = True, 0.3
    a, b = _render_skip_pair(photon, noisy, monkeypatch)
    assert np.array_equal(a, b)


def _morton_keys_numpy(x, y):
    """photon_sort.hip's keys in numpy: one scale for both axes over the range's bounding box, f64 arithmetic, truncation."""
    x, y = x.astype(np.float64), y.astype(np.float64)
    ext = max(x.max() - x.min(), y.max() - y.min())
    f = 65535.0 / ext if ext > 0 else 0.0
    ix, iy = ((x - x.min()) * f).astype(np.uint64) & 0xffff, ((y - y.min()) * f).astype(np.uint64) & 0xffff

    def spread(v):
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        return (v | (v << 1)) & 0x55555555
    return (spread(ix) | (spread(iy) << 1)).astype(np.uint32)


@pytest.mark.parametrize("n,first,count", [(1, 0, 1), (63, 0, 63), (4096, 0, 4096), (4097, 0, 4097), (100_003, 0, 100_003),
                                           (100_003, 777, 50_000), (1_000_000, 0, 1_000_000), (300_000, 0, 300_000)])
def test_morton_order_is_numpys_stable_argsort(photon, n, first, count):
    """The hand-written stable radix sort behind lens-major launches (photon_sort.hip: four passes of eight bits; histogram,
    scan, a scatter in which one wave ranks its tile with ballots) against numpy: the permutation is EXACTLY the stable
    argsort of the Morton keys -- ties (sources in one grid cell; here a fifth of the points are duplicates of others) keep the
    caller's order -- for sizes around the tile and wave boundaries, a sub-range, and C5's million."""
    rng = np.random.default_rng(n + first)
    x = rng.uniform(-3.0e4, 3.0e4, n).astype(np.float32)
    y = rng.uniform(-7.5e3, 9.0e3, n).astype(np.float32)               # a strip: one scale for both axes
    if n > 10:
        dup = rng.integers(0, n, n // 5)
        x[dup], y[dup] = x[(dup * 7 + 1) % n], y[(dup * 7 + 1) % n]     # exact ties
    if n == 300_000:
        x[:], y[:] = np.float32(12.5), np.float32(-3.0)                 # every key equal: the identity
    got = photon.morton_order(x, y, first, count)
    keys = _morton_keys_numpy(x[first:first + count], y[first:first + count])
    want = first + np.argsort(keys, kind="stable").astype(np.int32)
    assert np.array_equal(got, want), int(np.flatnonzero(got != want)[0])


def test_narrow_cones_keep_every_lens_sample(photon):
    """BOS (ray_cone_pitch_ratio 1e-4): every lens sample lands well inside the aperture; nothing is ruled out."""
    call = scenes.bos_scene(n_dots=5, points_per_dot=10, rays_per_source=64)
    scene = photon.scene_create(call)
    assert scene.live_rays() == 64
    scene.free()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_dead_lens_samples_against_exact_geometry(photon, seed):
    """The bound behind the static skip, held against the geometry itself in f64: for randomly placed fields (wide, deep, off
    centre) EVERY sample the scene rules out, from EVERY source, meets the thick lens's front sphere more than pitch / 2 from
    the axis (or misses it) -- by more than a micron: the kernels' f32 arithmetic is far inside that.  And the bound is not
    lazy: of the samples it keeps, those beyond 0.52 pitch are few."""
    rng = np.random.default_rng(seed)
    call = scenes.piv_scene(n_particles=400, rays_per_source=4000, mie=False, seed=seed,
                            field_half_width=float(rng.uniform(5e4, 4e5)))
    call.src_z = (call.src_z + rng.uniform(-1.5e5, 1.5e5, call.src_z.size)).astype(call.src_z.dtype)
    call.src_x = (call.src_x + rng.uniform(-1e5, 1e5)).astype(call.src_x.dtype)
    scene = photon.scene_create(call)
    live = scene.live_samples()
    scene.free()
    rps = call.lightray_number_per_particle
    dead = np.setdiff1d(np.arange(rps), live)
    assert 0.3 * rps < dead.size < 0.55 * rps
    r1, r2 = photon.rand_table(rps)
    e = call.elements[0]
    pitch = float(e["element_geometry"]["pitch"])
    Rf = float(e["element_geometry"]["front_surface_radius"])
    t = float(e["element_geometry"]["vertex_distance"])
    zc = float(call.element_center[0][2])
    ratio = float(call.ray_cone_pitch_ratio)
    lens_pitch = float(call.lens_pitch)
    za = float(call.image_distance)
    px = ratio * lens_pitch * r1.astype(np.float64) * np.cos(2 * np.pi * r2.astype(np.float64))
    py = ratio * lens_pitch * r1.astype(np.float64) * np.sin(2 * np.pi * r2.astype(np.float64))
    rad = np.hypot(px, py)
    assert (rad[live] > 0.52 * pitch).mean() < 0.08                    # nearly everything kept is inside, or just outside, the aperture
    sx, sy, sz = (np.asarray(a, np.float64)[:, None] for a in (call.src_x, call.src_y, call.src_z))
    # direction of ray (source, dead sample): through (px, py, za); the front sphere's centre (.cu:507-520)
    dx, dy, dz = px[dead][None, :] - sx, py[dead][None, :] - sy, za - sz
    cz = zc + (t / 2.0 - Rf)
    ox, oy, oz = sx, sy, sz - cz
    a = dx * dx + dy * dy + dz * dz
    b = 2 * (dx * ox + dy * oy + dz * oz)
    c = ox * ox + oy * oy + oz * oz - Rf * Rf
    disc = b * b - 4 * a * c
    hit = disc >= 0
    root = np.sqrt(np.where(hit, disc, 0.0))
    t1, t2 = (-b + root) / (2 * a), (-b - root) / (2 * a)
    tt = np.minimum(t1, t2) if Rf > 0 else np.maximum(t1, t2)          # .cu:298-336
    hx, hy = sx + tt * dx, sy + tt * dy
    rho = np.hypot(hx, hy)
    assert (~hit | (rho > pitch / 2.0 + 1.0)).all(), float(rho[hit].min() - pitch / 2.0)
