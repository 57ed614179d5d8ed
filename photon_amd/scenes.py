"""Synthetic workloads of BASELINE.json and the NRRD writer.

C0  PIV sample parameters, 1e4 rays, diffuse, no volume            (plumbing)
C2  PIV, 1e6 rays, thick lens, Mie table, 4-pixel splat
C3  BOS, 1e7 rays, 256^3 density volume, RK4, erf splat D=3        (headline)
C4  BOS, 1e8 rays, 512^3 volume (8 GPUs)
C5  Mie PIV + volume, 4e7 rays, 1e6 polydisperse sources (8 GPUs)

Every generator takes size knobs so the parity tests can run the same scene at sizes the CPU
oracle finishes in seconds.  Scene set-up mirrors what the reference's driver does before the hot
path (python_codes/run_simulation_02.py:774-996 PIV particles, :999-1056 + :1328-1551 BOS dots)
without copying it: these are synthetic stand-ins with fixed seeds, not its exact patterns.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np

from .ray_tracing import RayTracingCall, single_lens_camera

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# photon's sample camera (sample-data/*/parameters/sample-parameters.mat)
SAMPLE_LENS = dict(focal_length=105000.0, aperture_f_number=8.0, object_distance=700000.0,
                   lens_radius_of_curvature=100000.0)


def sample_camera(implement_diffraction: bool, n_pixels: int = 1024, pixel_pitch: float = 17.0,
                  diffraction_diameter: float = 3.0) -> dict:
    return dict(pixel_bit_depth=10, pixel_gain=25.0, pixel_pitch=pixel_pitch, x_camera_angle=0.0, y_camera_angle=0.0,
                x_pixel_number=n_pixels, y_pixel_number=n_pixels, z_sensor=0.0,
                diffraction_diameter=diffraction_diameter, implement_diffraction=implement_diffraction,
                rotation_matrix=np.eye(3), inverse_rotation_matrix=np.eye(3))


# ----------------------------------------------------------------------------------------------
# NRRD (what loadNRRD reads: trace_rays_through_density_gradients.h:1663-1817)
# ----------------------------------------------------------------------------------------------


def write_nrrd(path: str, rho: np.ndarray, spacing, origin) -> str:
    """rho[z, y, x] float32 -> NRRD0005, raw little-endian, x fastest."""
    rho = np.ascontiguousarray(rho, dtype="<f4")
    nz, ny, nx = rho.shape
    header = ("NRRD0005\n# written by photon_amd.scenes.write_nrrd\ntype: float\ndimension: 3\n"
              "space: 3D-left-handed\n"
              f"sizes: {nx} {ny} {nz}\nendian: little\nencoding: raw\n"
              f"spacings: {spacing[0]!r} {spacing[1]!r} {spacing[2]!r}\n"
              f"space origin: ({origin[0]!r},{origin[1]!r},{origin[2]!r})\n\n")
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(rho.tobytes())
    return path


def read_nrrd(path: str):
    """Minimal reader for the files write_nrrd / pynrrd produce (tests only)."""
    with open(path, "rb") as f:
        fields = {}
        assert f.readline().startswith(b"NRRD")
        while True:
            line = f.readline().decode("ascii", "replace").rstrip("\r\n")
            if line == "":
                break
            if line.startswith("#") or ":" not in line:
                continue
            k, v = line.split(":", 1)
            fields[k.strip()] = v.strip().lstrip("=")
        nx, ny, nz = (int(t) for t in fields["sizes"].split())
        rho = np.frombuffer(f.read(nx * ny * nz * 4), dtype="<f4").reshape(nz, ny, nx)
    spacing = [float(t) for t in fields.get("spacings", "1 1 1").split()]
    origin = [float(t) for t in fields.get("space origin", "(0,0,0)").strip("()").split(",")]
    return rho, spacing, origin


def gaussian_blob_density(n: int, spacing: float, origin, sigma: float = 8.0e3, amplitude: float = 0.2,
                          rho0: float = 1.225, dtype=np.float32) -> np.ndarray:
    """rho = rho0 + A exp(-|r - rc|^2 / (2 sigma^2)), rc = centre of the volume."""
    ax = [origin[a] + spacing * np.arange(n, dtype=np.float64) for a in range(3)]
    c = [origin[a] + spacing * (n - 1) / 2.0 for a in range(3)]
    gx = np.exp(-((ax[0] - c[0]) ** 2) / (2 * sigma ** 2))
    gy = np.exp(-((ax[1] - c[1]) ** 2) / (2 * sigma ** 2))
    gz = np.exp(-((ax[2] - c[2]) ** 2) / (2 * sigma ** 2))
    out = np.empty((n, n, n), dtype=dtype)
    for k in range(n):                     # slab by slab: 512^3 in float64 would not fit comfortably
        out[k] = (rho0 + amplitude * gz[k] * np.outer(gy, gx)).astype(dtype)
    return out


def bos_volume(n: int = 256, extent: float = 66300.0, origin_z: float = 300000.0):
    """Synthetic BOS volume: n^3 grid spanning `extent` microns, centred on the optical axis,
    placed between lens and target (the shipped sample volume is missed by
    every ray).  Returns (rho[z,y,x], spacing[3], origin[3])."""
    spacing = extent / (n - 1)
    origin = (-extent / 2.0, -extent / 2.0, origin_z)
    return gaussian_blob_density(n, spacing, origin, sigma=8.0e3 * extent / 66300.0), (spacing,) * 3, origin


# ----------------------------------------------------------------------------------------------
# sources
# ----------------------------------------------------------------------------------------------


def sunflower_disc(n_points: int, diameter: float) -> np.ndarray:
    """Vogel spiral: n points filling a disc uniformly -> [n, 2]."""
    k = np.arange(n_points, dtype=np.float64) + 0.5
    r = 0.5 * diameter * np.sqrt(k / n_points)
    t = k * np.pi * (3.0 - np.sqrt(5.0))
    return np.stack([r * np.cos(t), r * np.sin(t)], 1)


def concentric_disc(diameter: float, n_points: float, phases) -> np.ndarray:
    """The reference's BOS dot template (calculate_sunflower_coordinates, run_simulation_02.py:999-1056): points on
    concentric circles one nearest-neighbour spacing apart, each circle turned by 2 pi * phases[k] (the reference
    draws that phase from numpy's global generator), plus the centre -> [n, 2].  Note the reference's own quirk:
    a circle meant to carry m points gets m - 1 (np.arange(1, m) - 1).  Pinned by tests/golden/pins.npz."""
    area = np.pi * (diameter / 2.0) ** 2.0
    spacing = np.sqrt(area / n_points)
    radii = np.linspace(spacing, diameter / 2.0, int(np.round((diameter / 2.0) / spacing)))
    rho = 1.0 / spacing
    xs, ys = [], []
    for k, r in enumerate(radii):
        m = np.round(rho * (2.0 * np.pi * r))
        theta = (2.0 * np.pi / m) * (np.arange(1.0, m) - 1.0) + 2.0 * np.pi * float(phases[k])
        xs.append(r * np.cos(theta))
        ys.append(r * np.sin(theta))
    return np.stack([np.append(np.concatenate(xs), 0.0), np.append(np.concatenate(ys), 0.0)], 1)


def _call(cam_geom, camera, **kw) -> RayTracingCall:
    return RayTracingCall(
        lens_pitch=cam_geom["lens_pitch"], image_distance=cam_geom["image_distance"],
        z_offset=cam_geom["z_offset"], object_distance=cam_geom["object_distance"],
        aperture_f_number=SAMPLE_LENS["aperture_f_number"], elements=[cam_geom["element"]],
        element_center=cam_geom["element_center"], element_plane_parameters=cam_geom["element_plane_parameters"],
        element_system_index=cam_geom["element_system_index"], camera=camera, source_point_number=10000, **kw)


def bos_pattern(n_dots: int = 200, points_per_dot: int = 100, field_half_width: float = 3.0e4,
                dot_diameter: float = 600.0, seed: int = 1):
    """Dot centres [n_dots, 2] and the per-dot point template [points_per_dot, 2] of bos_scene: the inputs of the
    on-device generator photon_sources_bos (source g*P + j = centre g + template j)."""
    rng = np.random.default_rng(seed)
    return rng.uniform(-field_half_width, field_half_width, size=(n_dots, 2)), sunflower_disc(points_per_dot, dot_diameter)


def bos_scene(n_dots: int = 200, points_per_dot: int = 100, rays_per_source: int = 500,
              density_grad_filename: str = "", field_half_width: float = 3.0e4, dot_diameter: float = 600.0,
              seed: int = 1, lens_model: str = "general", n_pixels: int = 1024,
              ray_tracing_algorithm: int = 2) -> RayTracingCall:
    """C3 / C4: dot pattern at the object plane, diffuse sources, erf splat (D = 3 px)."""
    geom = single_lens_camera(lens_model=lens_model, **SAMPLE_LENS)
    centres, disc = bos_pattern(n_dots, points_per_dot, field_half_width, dot_diameter, seed)
    xy = (centres[:, None, :] + disc[None, :, :]).reshape(-1, 2)
    n = xy.shape[0]
    return _call(geom, sample_camera(True, n_pixels), scattering_type="diffuse",
                 src_x=xy[:, 0], src_y=xy[:, 1], src_z=np.full(n, geom["z_object"]),
                 src_radiance=np.full(n, 10.0), src_diameter_index=np.ones(n, np.int32),
                 lightray_number_per_particle=rays_per_source, beam_wavelength=0.0, ray_cone_pitch_ratio=1e-4,
                 simulate_density_gradients=bool(density_grad_filename), density_grad_filename=density_grad_filename,
                 ray_tracing_algorithm=ray_tracing_algorithm if density_grad_filename else 0)


def load_mie_table():
    """Mie table captured from the reference's create_mie_scattering_data
    (tests/golden/abi_piv.npz): angles [255], irradiance [255, 27]."""
    import json
    a = np.load(os.path.join(GOLDEN_DIR, "abi_piv.npz"))
    with open(os.path.join(GOLDEN_DIR, "abi_piv.json")) as f:
        j = json.load(f)
    na, nd = j["scattering"]["num_angles"], j["scattering"]["num_diameters"]
    return a["scattering_angle"].copy(), a["scattering_irradiance"].reshape(na, nd).copy()


def piv_scene(n_particles: int = 100, rays_per_source: int = 10000, mie: bool = True, seed: int = 0,
              density_grad_filename: str = "", polydisperse: bool = False, ray_cone_pitch_ratio: float = 1.0,
              n_pixels: int = 1024, ray_tracing_algorithm: int = 2, field_half_width: float = 7.5e4,
              sort_by_tile: bool = False) -> RayTracingCall:
    """C0 / C2 / C5: particles in a laser sheet, thick lens, 4-pixel splat."""
    geom = single_lens_camera(lens_model="general", **SAMPLE_LENS)
    rng = np.random.default_rng(seed)
    x = rng.uniform(-field_half_width, field_half_width, n_particles)
    y = rng.uniform(-field_half_width, field_half_width, n_particles)
    zl = rng.uniform(-7.5e3, 7.5e3, n_particles)
    sigma = 730.0 / (2.0 * np.sqrt(2.0 * np.log(2.0)))
    const = 500.0 if mie else 1.0e4
    radiance = const / (sigma * np.sqrt(2 * np.pi)) * np.exp(-zl ** 2 / (2 * sigma ** 2))
    kw = {}
    if mie:
        ang, irr = load_mie_table()
        kw = dict(scattering=dict(inverse_rotation_matrix=np.eye(3).reshape(9), beam_propagation_vector=[0.0, 1.0, 0.0]),
                  scattering_angle=ang, scattering_irradiance=irr)
        if polydisperse:        # log-normal-ish spread over the table's 27 diameters (C5)
            dia = np.clip(np.rint(rng.lognormal(np.log(13.0), 0.25, n_particles)), 0, irr.shape[1] - 1).astype(np.int32)
        else:                   # the reference always passes index 1 (run_simulation_02.py:992)
            dia = np.ones(n_particles, np.int32)
    else:
        dia = np.ones(n_particles, np.int32)
    if sort_by_tile:            # optional locality ordering of sources (coarse xy tiles)
        key = (np.floor((y + field_half_width) / 2000.0) * 4096 + np.floor((x + field_half_width) / 2000.0))
        o = np.argsort(key, kind="stable")
        x, y, zl, radiance, dia = x[o], y[o], zl[o], radiance[o], dia[o]
    return _call(geom, sample_camera(False, n_pixels), scattering_type="mie" if mie else "diffuse",
                 src_x=x, src_y=y, src_z=zl + geom["z_object"], src_radiance=radiance, src_diameter_index=dia,
                 lightray_number_per_particle=rays_per_source, beam_wavelength=0.532,
                 ray_cone_pitch_ratio=ray_cone_pitch_ratio,
                 simulate_density_gradients=bool(density_grad_filename), density_grad_filename=density_grad_filename,
                 ray_tracing_algorithm=ray_tracing_algorithm if density_grad_filename else 0, **kw)


def config(name: str, workdir: Optional[str] = None, scale: float = 1.0, volume_n: Optional[int] = None) -> RayTracingCall:
    """BASELINE.json configs by name ('C0','C2','C3','C4','C5'); `scale` shrinks ray counts and
    (for volumes) the grid so tests can run the same scene small."""
    name = name.upper()
    if name == "C0":
        return piv_scene(n_particles=100, rays_per_source=100, mie=False)
    if name == "C2":
        return piv_scene(n_particles=max(1, int(100 * scale)), rays_per_source=10000, mie=True)
    if name in ("C3", "C4", "C5"):
        assert workdir is not None, "volume configs need a directory for the NRRD file"
        n = {"C3": 256, "C4": 512, "C5": 256}[name]
        n = max(16, int(round(n * min(1.0, scale ** (1 / 3))))) if scale < 1 else n
        if volume_n is not None:        # e.g. one GPU's share of C4: fewer sources, full-size grid
            n = int(volume_n)
        path = os.path.join(workdir, f"bos_{n}.nrrd")
        if not os.path.exists(path):
            rho, sp, org = bos_volume(n)
            write_nrrd(path, rho, sp, org)
        if name == "C3":
            return bos_scene(n_dots=max(1, int(200 * scale)), density_grad_filename=path)
        if name == "C4":
            return bos_scene(n_dots=max(1, int(2000 * scale)), density_grad_filename=path)
        return piv_scene(n_particles=max(1, int(1_000_000 * scale)), rays_per_source=40, mie=True, polydisperse=True,
                         density_grad_filename=path, field_half_width=3.0e4, sort_by_tile=True)
    raise ValueError(name)
