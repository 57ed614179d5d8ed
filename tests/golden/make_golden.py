#!/usr/bin/env python3
"""Generate golden fixtures from the *reference's own Python* (this container only).

Runs photon's unmodified driver (`python_codes/run_simulation_02.py` ->
`perform_ray_tracing_03.py`) on the shipped sample PIV / BOS parameter files up to
the `ctypes.CDLL(...).start_ray_tracing(...)` call, which is intercepted by a
recorder.  What is captured (all DATA, no reference source text):

  abi_<case>.npz / abi_<case>.json
      every argument the reference's marshalling code
      (perform_ray_tracing_03.py:1631-1938) hands to `start_ray_tracing`, structs
      flattened field by field  -> pins photon_amd.ray_tracing (our mirror of the
      marshalling layer) and gives the parity tests real sample-data inputs.
  postprocess_<case>.npz
      a synthetic raw sensor image written by the recorder + the uint16 image the
      reference's post-processing (perform_ray_tracing_03.py:2190-2247) made of it
      -> pins photon_amd.ray_tracing.postprocess_image.
  lens_f64.npz
      inputs/outputs of the reference's float64 numpy ancestors of the device
      functions: ray_sphere_intersection (perform_ray_tracing_03.py:472),
      measure_distance_to_optical_axis (:585), propogate_rays_through_single_element
      (:671)  -> loose (f32-vs-f64) pin of oracle rows a15/a16/a17.
  mie_table.npz
      Mie table + angles from create_mie_scattering_data (run_simulation_02.py:699).

/root/reference is never read at test time; only these files travel.
Usage:  python tests/golden/make_golden.py            (needs /root/reference)
"""
import ctypes
import json
import os
import shutil
import sys
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True

# py2 / numpy-1 era names the reference uses
np.NAN = np.nan
np.infty = np.inf
np.object = object


def _ptr_value(p):
    if p is None:
        return 0
    if isinstance(p, int):
        return p
    v = getattr(p, "value", None)
    if v is not None:
        return int(v)
    return int(ctypes.cast(p, ctypes.c_void_p).value or 0)


def _read(addr, ctype, n):
    if not addr or n <= 0:
        return np.zeros(0, dtype=np.dtype(ctype))
    buf = (ctype * n).from_address(addr)
    return np.array(buf, copy=True)


def _struct_scalars(s, skip=()):
    out = {}
    for name, _typ in s._fields_:
        if name in skip:
            continue
        v = getattr(s, name)
        if isinstance(v, ctypes.Structure):
            out[name] = _struct_scalars(v)
        elif isinstance(v, ctypes.Array):
            out[name] = [float(x) for x in v]
        elif isinstance(v, bytes):
            out[name] = v.decode()
        elif isinstance(v, (bool, int, float)):
            out[name] = v
        else:
            out[name] = _ptr_value(v)
    return out


class Recorder:
    """Stands in for the loaded .so; records one call, writes a synthetic image."""

    def __init__(self):
        self.calls = []
        self.argtypes = None
        self.restype = None

    def __call__(self, *a):
        (lens_pitch, image_distance, scat, scat_type, src, rays_per_src, wavelength,
         f_number, num_elements, elem_center, elem_data, elem_plane, elem_sys_idx, cam,
         image, sim_grad, grad_file, save_rays, pos_path, dir_path, num_save, algo,
         add_pos_noise, pos_noise_std, add_ngrad_noise, ngrad_noise_std, cone_ratio,
         save_inter, num_inter) = a
        n_src = int(src.num_particles)
        rec = {
            "scalars": dict(
                lens_pitch=float(lens_pitch), image_distance=float(image_distance),
                scattering_type=scat_type.decode(), lightray_number_per_particle=int(rays_per_src),
                beam_wavelength=float(wavelength), aperture_f_number=float(f_number),
                num_elements=int(num_elements), simulate_density_gradients=bool(sim_grad),
                density_grad_filename=os.path.basename(grad_file.decode()),
                save_lightrays=bool(save_rays), num_lightrays_save=int(num_save),
                ray_tracing_algorithm=int(algo), add_pos_noise=bool(add_pos_noise),
                pos_noise_std=float(pos_noise_std), add_ngrad_noise=bool(add_ngrad_noise),
                ngrad_noise_std=float(ngrad_noise_std), ray_cone_pitch_ratio=float(cone_ratio),
                save_intermediate_ray_data=bool(save_inter),
                num_intermediate_positions_save=int(num_inter)),
            "scattering": _struct_scalars(scat, skip=("scattering_angle", "scattering_irradiance")),
            "source": _struct_scalars(src, skip=("diameter_index", "radiance", "x", "y", "z")),
            "camera": _struct_scalars(cam),
            "elements": [_struct_scalars(elem_data[i]) for i in range(int(num_elements))],
            "sizeof": dict(scattering=ctypes.sizeof(scat), source=ctypes.sizeof(src),
                           camera=ctypes.sizeof(cam), element=ctypes.sizeof(elem_data[0])),
        }
        arrays = dict(
            src_x=_read(_ptr_value(src.x), ctypes.c_float, n_src),
            src_y=_read(_ptr_value(src.y), ctypes.c_float, n_src),
            src_z=_read(_ptr_value(src.z), ctypes.c_float, n_src),
            src_radiance=_read(_ptr_value(src.radiance), ctypes.c_double, n_src),
            src_diameter_index=_read(_ptr_value(src.diameter_index), ctypes.c_int, n_src),
            element_center=np.array(elem_center, dtype=np.float64, copy=True),
            element_plane_parameters=np.array(elem_plane, dtype=np.float64, copy=True),
            element_system_index=np.array(elem_sys_idx, dtype=np.int32, copy=True),
            scattering_angle=_read(_ptr_value(scat.scattering_angle), ctypes.c_float,
                                   int(scat.num_angles)),
            scattering_irradiance=_read(_ptr_value(scat.scattering_irradiance), ctypes.c_float,
                                        int(scat.num_angles) * int(scat.num_diameters)),
            # raw bytes of the structs exactly as the callee would see them
            raw_camera=np.frombuffer(bytes(cam), dtype=np.uint8).copy(),
            raw_element0=np.frombuffer(bytes(elem_data[0]), dtype=np.uint8).copy(),
        )
        # deterministic synthetic "rendered" image so the reference's post-processing
        # has something to chew on (sparse blobs + one NaN + one negative)
        rng = np.random.default_rng(20240607 + len(self.calls))
        img = np.zeros(image.size, dtype=np.float32)
        idx = rng.integers(0, image.size, size=5000)
        img[idx] = rng.gamma(2.0, 3.0, size=idx.size).astype(np.float32)
        img[7] = -1.5
        image.reshape(-1)[:] = img
        arrays["synthetic_image"] = img.copy()
        self.calls.append((rec, arrays))


def run_case(case, shrink):
    import scipy.io as sio
    import helper_functions
    import run_simulation_02 as rs
    import perform_ray_tracing_03 as prt
    prt.long = int
    import builtins
    import bhmie as _bh           # `from numpy import *` shadows builtins under numpy 2
    _bh.max, _bh.min = builtins.max, builtins.min

    work = tempfile.mkdtemp(prefix="photon_golden_")
    os.makedirs(os.path.join(work, "python_codes"))
    os.makedirs(os.path.join(work, "cuda_codes"))
    os.symlink(os.path.join(REF, "sample-data"), os.path.join(work, "sample-data"))
    os.chdir(os.path.join(work, "python_codes"))
    os.environ.setdefault("LD_LIBRARY_PATH", "")

    rec = Recorder()

    class FakeLib:
        start_ray_tracing = rec

    prt.ctypes.CDLL = lambda *_a, **_k: FakeLib
    # the PIV branch chdir()s into the (read-only) reference tree; keep us where we are
    real_chdir = os.chdir
    os.chdir = lambda d: None if os.path.realpath(d).startswith(REF) else real_chdir(d)

    post = []
    orig = rs.perform_ray_tracing_03

    def wrapped(*a, **k):
        I, I_raw = orig(*a, **k)
        post.append((np.array(I, copy=True), np.array(I_raw, copy=True)))
        return I, I_raw

    rs.perform_ray_tracing_03 = wrapped

    # --- what batch_run_simulation.py:71-115 does, minus argv ---
    m = sio.loadmat(os.path.join(REF, "sample-data", case, "parameters", "sample-parameters.mat"),
                    struct_as_record=False, squeeze_me=True)
    p = {k: v for k, v in m.items() if not k.startswith("__")}
    for k in list(p):
        if hasattr(p[k], "_fieldnames"):
            p[k] = helper_functions._todict(p[k])
    p["output_data"]["image_directory"] = os.path.join(work, "images")
    for i in p:
        if isinstance(p[i], (bytes, list, str)):
            continue
        for j in p[i]:
            if type(p[i][j]) is int:
                p[i][j] = float(p[i][j])
    shrink(p)
    rs.run_simulation_02(p)
    rs.perform_ray_tracing_03 = orig
    os.chdir = real_chdir
    os.chdir(OUT)
    shutil.rmtree(work, ignore_errors=True)
    return rec.calls, post, p


def save_case(name, call, post, params):
    rec, arrays = call
    cd = params["camera_design"]
    rec["postprocess"] = dict(pixel_gain=float(cd["pixel_gain"]),
                              pixel_bit_depth=float(cd["pixel_bit_depth"]),
                              image_noise=float(cd["image_noise"]),
                              intensity_rescaling=bool(cd.get("intensity_rescaling", True)))
    rec["lens_design"] = {k: (float(v) if not isinstance(v, str) else v)
                          for k, v in params["lens_design"].items()}
    with open(os.path.join(OUT, f"abi_{name}.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True, default=float)
    synth = arrays.pop("synthetic_image")
    np.savez_compressed(os.path.join(OUT, f"abi_{name}.npz"), **arrays)
    I, I_raw = post
    nz = np.flatnonzero(synth)
    np.savez_compressed(os.path.join(OUT, f"postprocess_{name}.npz"),
                        raw_index=nz.astype(np.int64), raw_value=synth[nz],
                        shape=np.array(I.shape), out_u16_index=np.flatnonzero(I).astype(np.int64),
                        out_u16_value=I.reshape(-1)[np.flatnonzero(I)])


def lens_golden():
    import run_simulation_02  # noqa: F401  (resolves the circular import)
    import perform_ray_tracing_03 as prt
    prt.long = int
    rng = np.random.default_rng(7)
    n = 4096
    # --- ray_sphere_intersection: sample-lens geometry, both surfaces ---
    zc_lens = 123598.87980659823
    t, R = 431.12874248149456, 1.0e5
    out = {}
    for tag, Rs, cz in (("front", +R, zc_lens + t / 2 - R), ("back", -R, zc_lens - t / 2 + R)):
        x0 = rng.uniform(-7.5e4, 7.5e4, n)
        y0 = rng.uniform(-7.5e4, 7.5e4, n)
        z0 = np.full(n, 823668.3478484906) + rng.uniform(-7.5e3, 7.5e3, n)
        lx = rng.uniform(-9e3, 9e3, n)
        ly = rng.uniform(-9e3, 9e3, n)
        v = np.stack([lx - x0, ly - y0, zc_lens - z0], 1)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        if tag == "back":   # start just inside the glass
            z0 = np.full(n, zc_lens + 100.0)
            x0, y0 = lx, ly
        xi, yi, zi = prt.ray_sphere_intersection(0.0, 0.0, cz, Rs, v[:, 0], v[:, 1], v[:, 2],
                                                 x0, y0, z0, tag)
        out[f"rsi_{tag}_in"] = np.stack([x0, y0, z0, v[:, 0], v[:, 1], v[:, 2]], 1)
        out[f"rsi_{tag}_c"] = np.array([0.0, 0.0, cz, Rs])
        out[f"rsi_{tag}_out"] = np.stack([xi, yi, zi], 1)
    # --- measure_distance_to_optical_axis ---
    pts = rng.uniform(-2e4, 2e4, (n, 3)) + np.array([0, 0, zc_lens])
    lens_center = np.array([[0.0, 0.0, zc_lens]])
    plane = np.array([[0.0, 0.0, 1.0, -zc_lens]])
    d = prt.measure_distance_to_optical_axis(pts[:, 0], pts[:, 1], pts[:, 2], lens_center.T, plane.T)
    out["axis_in"] = pts
    out["axis_out"] = np.squeeze(np.asarray(d))
    # --- full thick-lens element ---
    elem = dict(element_type="lens",
                element_geometry=dict(pitch=13125.0, vertex_distance=t, front_surface_radius=R,
                                      back_surface_radius=-R),
                element_properties=dict(refractive_index=1.476521991017817, abbe_number=np.nan,
                                        transmission_ratio=1.0, absorbance_rate=0.0))
    x0 = rng.uniform(-7.5e4, 7.5e4, n)
    y0 = rng.uniform(-7.5e4, 7.5e4, n)
    z0 = 823668.3478484906 + rng.uniform(-7.5e3, 7.5e3, n)
    rr = 13125.0 * rng.uniform(0, 1, n)          # CUDA-style: radius up to a full pitch
    ph = 2 * np.pi * rng.uniform(0, 1, n)
    v = np.stack([rr * np.cos(ph) - x0, rr * np.sin(ph) - y0, 123529.41176470589 - z0], 1)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    lrd = dict(ray_source_coordinates=np.stack([x0, y0, z0], 1).copy(),
               ray_propogation_direction=v.copy(),
               ray_wavelength=np.full(n, 0.532), ray_radiance=np.ones(n))
    out["lens_in"] = np.concatenate([lrd["ray_source_coordinates"], v], 1)
    res = prt.propogate_rays_through_single_element(elem, lens_center[0], plane[0], lrd)
    out["lens_out"] = np.concatenate([res["ray_source_coordinates"],
                                      res["ray_propogation_direction"]], 1)
    out["lens_out_radiance"] = np.asarray(res["ray_radiance"])
    np.savez_compressed(os.path.join(OUT, "lens_f64.npz"), **out)


def main():
    sys.path.insert(0, os.path.join(REF, "python_codes"))

    def shrink_piv(p):
        p["particle_field"]["particle_number"] = 300.0
        p["particle_field"]["lightray_number_per_particle"] = 100.0
        p["particle_field"]["frame_vector"] = np.array([1])

    def shrink_bos(p):
        p["bos_pattern"]["grid_point_number"] = 40
        p["bos_pattern"]["particle_number_per_grid_point"] = 25

    calls, post, params = run_case("piv", shrink_piv)
    save_case("piv", calls[0], post[0], params)
    calls, post, params = run_case("bos", shrink_bos)
    save_case("bos_im1", calls[0], post[0], params)
    save_case("bos_im2", calls[1], post[1], params)
    lens_golden()
    # the one sample volume the reference ships (data, 1 MiB)
    shutil.copyfile(os.path.join(REF, "sample-data", "bos", "sample-density.nrrd"),
                    os.path.join(OUT, "sample-density.nrrd"))
    os.chmod(os.path.join(OUT, "sample-density.nrrd"), 0o644)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
