import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (runs through the HIP library)")


def _gpu_available() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def photon():
    """The HIP product library.  On a box with a GPU device node the GPU tests FAIL (not skip) when the library
    cannot be loaded or the runtime sees no device; they skip only where there is no GPU at all."""
    from photon_amd.library import PhotonLibrary
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU in this environment (/dev/kfd absent)")
    # torch first: several tests hand torch device tensors to the library, and a process must run ONE HIP runtime.
    # torch bundles its own libamdhip64; loaded first, the library (RUNPATH /opt/rocm/lib) binds to that copy by
    # SONAME instead of mapping the system's next to it (two runtimes in one process: torch then finds no device).
    import torch
    assert torch.cuda.is_available(), "a GPU device node exists but torch sees no device"
    lib = PhotonLibrary()
    lib.set_device(0)               # raises if HIP cannot reach the device: a broken runtime must not go green
    return lib


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("photon"))


def load_fixture_call(name: str):
    from photon_amd.ray_tracing import RayTracingCall
    return RayTracingCall.from_fixture(os.path.join(GOLDEN, f"abi_{name}.json"), os.path.join(GOLDEN, f"abi_{name}.npz"),
                                       density_dir=GOLDEN)


# ---- the BOS image pair with ray dumps that pins the dump wire format to the reference's own reader -------------------
DUMP_SLOTS = 6


def dump_pair_calls(workdir: str):
    """Deterministic small BOS pair (image 1 without, image 2 with the density-gradient volume) that writes every ray
    dump the reference knows: pos_/dir_ per image, intermediate_pos_/intermediate_dir_ for image 2, in the folder
    layout its reader expects (light_ray_processing.py:160-207).  Used by tests/golden/make_golden.py (oracle writes,
    the REFERENCE's reader parses -> dumps_reference_reader.npz) and by the tests (oracle / GPU write, our parse)."""
    from photon_amd import scenes
    rho, sp, org = scenes.bos_volume(40)
    nrrd = scenes.write_nrrd(os.path.join(workdir, "dump_pair_40.nrrd"), rho, sp, org)
    calls = []
    for im, grad in (("im1", False), ("im2", True)):
        c = scenes.bos_scene(n_dots=3, points_per_dot=8, rays_per_source=25, density_grad_filename=nrrd if grad else "", seed=5)
        c.simulate_density_gradients = grad
        c.ray_tracing_algorithm = 2 if grad else 0
        c.save_lightrays, c.num_lightrays_save = True, c.num_rays
        c.save_intermediate_ray_data, c.num_intermediate_positions_save = grad, DUMP_SLOTS if grad else 0
        pdir = os.path.join(workdir, "light-ray-positions", im)
        ddir = os.path.join(workdir, "light-ray-directions", im)
        os.makedirs(pdir, exist_ok=True)
        os.makedirs(ddir, exist_ok=True)
        c.lightray_position_save_path, c.lightray_direction_save_path = pdir, ddir
        calls.append(c)
    return calls


def parse_dump_pair(folder: str):
    """Our reading of the dump wire format (include/parallel_ray_tracing.h): flat little-endian f32 triples, ray-major;
    intermediates [ray][slot][3]; the reference's reader reports directions as arccos of the components."""
    import numpy as np
    out = {}
    for im in ("im1", "im2"):
        p = np.fromfile(os.path.join(folder, "light-ray-positions", im, "pos_0000.bin"), np.float32).reshape(-1, 3)
        d = np.fromfile(os.path.join(folder, "light-ray-directions", im, "dir_0000.bin"), np.float32).reshape(-1, 3)
        out[f"pos_{im}"], out[f"ang_{im}"] = p, np.arccos(d)
    ip = np.fromfile(os.path.join(folder, "light-ray-positions", "im2", "intermediate_pos_0000.bin"), np.float32)
    idr = np.fromfile(os.path.join(folder, "light-ray-directions", "im2", "intermediate_dir_0000.bin"), np.float32)
    out["ipos"] = ip.reshape(-1, DUMP_SLOTS, 3)
    out["iang"] = np.arccos(idr.reshape(-1, DUMP_SLOTS, 3))
    return out


# ---- multi-element trains pinned to the reference's own numpy sequencer ------------------------------------------------
def train_cases():
    """Trains of single-member groups -- the branch of the reference's numpy sequencer that runs
    (propogate_rays_through_optical_system, perform_ray_tracing_03.py:1419-1485) -- on a small volume-free PIV scene.
    Returns {name: RayTracingCall}.  Used by tests/golden/make_golden.py::train_golden (the oracle generates the rays,
    the REFERENCE propagates them in float64 -> train_f64.npz) and by the CPU / GPU tests.  `a_fwd` and `a_swap` are
    the same two lenses listed in either array order with the system indices swapped along (the sequencer visits
    DECREASING system index, :1421-1422, so both are the same physical train); `a_rev` sends the rays through the
    sensor-side lens first; `d_lens_stop` ends in an aperture (the reference's numpy leaves a ray that passes a stop
    where it was, the device code moves it to the stop's back plane, .cu:985-1003: same line, so the tests compare
    lines and sensor hits, not the point along the line)."""
    import copy
    import numpy as np
    from photon_amd import scenes
    base = scenes.piv_scene(n_particles=24, rays_per_source=96, mie=False, seed=11, field_half_width=2.5e4,
                            ray_cone_pitch_ratio=0.7)
    zc = float(base.element_center[0][2])
    pitch = float(base.elements[0]["element_geometry"]["pitch"])

    def lens(rf, rb, n, t):
        e = copy.deepcopy(base.elements[0])
        e["element_type"] = "l"
        e["element_geometry"].update(front_surface_radius=rf, back_surface_radius=rb, vertex_distance=t, pitch=pitch)
        e["element_properties"].update(refractive_index=n, transmission_ratio=0.97)
        return e

    def aperture(p, t):
        e = copy.deepcopy(base.elements[0])
        e["element_type"] = "a"
        e["element_geometry"].update(pitch=p, vertex_distance=t)
        return e

    A, zA = lens(1.6e5, -2.2e5, 1.52, 400.0), zc + 1500.0
    B, zB = lens(2.5e5, -1.3e5, 1.476, 450.0), zc - 1500.0
    trains = {
        "a_fwd": ([A, B], [zA, zB], [2, 1]),
        "a_swap": ([B, A], [zB, zA], [1, 2]),
        "a_rev": ([A, B], [zA, zB], [1, 2]),
        "b_stop_lens": ([aperture(9000.0, 50.0), copy.deepcopy(base.elements[0])], [zc + 4000.0, zc], [2, 1]),
        "c_three": ([A, aperture(8000.0, 20.0), B], [zA, zc, zB], [3, 2, 1]),
        "d_lens_stop": ([copy.deepcopy(base.elements[0]), aperture(7000.0, 30.0)], [zc, zc - 3000.0], [2, 1]),
    }
    out = {}
    for name, (elems, zs, idx) in trains.items():
        c = copy.deepcopy(base)
        c.elements = [copy.deepcopy(e) for e in elems]
        c.element_center = np.array([[0.0, 0.0, z] for z in zs])
        c.element_plane_parameters = np.array([[0.0, 0.0, 1.0, -z] for z in zs])
        c.element_system_index = np.array(idx, np.int32)
        out[name] = c
    return out


# ---- end-to-end BOS displacement: constant density gradient between target and lens ----------------------------------
def bos_displacement_case(workdir: str, target_px: float = 3.0):
    """A BOS dot pattern rendered without (call 1) and through (call 2) a volume of constant d(rho)/dx sized so that the
    reference's own paraxial relation (python_codes/nrrd_functions.py:60-82: epsilon = n_grad * del_z / n_0,
    displacement = M * Z_D * epsilon / pixel_pitch) predicts a dot shift of `target_px` pixels.  Returns
    (call_without, call_with, predicted_shift_px)."""
    import numpy as np
    from photon_amd import scenes
    from photon_amd.ray_tracing import single_lens_camera
    n, extent, pitch, K, rho_0 = 64, 66300.0, 17.0, 0.225e-3, 1.225
    h = extent / (n - 1)
    geom = single_lens_camera(lens_model="general", **scenes.SAMPLE_LENS)
    M = scenes.SAMPLE_LENS["focal_length"] / (scenes.SAMPLE_LENS["object_distance"] - scenes.SAMPLE_LENS["focal_length"])
    origin = (-extent / 2, -extent / 2, 300000.0)
    Z_D = geom["z_object"] - (origin[2] + geom["z_offset"] + extent / 2)      # dot pattern to the middle of the volume
    n_0 = K * rho_0 + 1
    epsilon = target_px * pitch / (M * Z_D)
    rho_grad = epsilon * n_0 / extent / K                                     # density per micron along x
    rho = np.broadcast_to((rho_0 + rho_grad * np.arange(n) * h)[None, None, :], (n, n, n)).astype(np.float32)
    nrrd = scenes.write_nrrd(os.path.join(workdir, "linear_density_64.nrrd"), rho, (h, h, h), origin)
    kw = dict(n_dots=6, points_per_dot=30, rays_per_source=100, seed=7, field_half_width=1.5e4)
    predicted = M * Z_D * (1 / n_0 * (K * rho_grad) * extent) / pitch          # the reference's formula, evaluated forward
    return scenes.bos_scene(**kw), scenes.bos_scene(density_grad_filename=nrrd, **kw), predicted


def image_centroid(img):
    import numpy as np
    img = np.asarray(img, np.float64)
    rows, cols = np.indices(img.shape)
    return (img * cols).sum() / img.sum(), (img * rows).sum() / img.sum()
