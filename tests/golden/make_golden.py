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
  train_f64.npz
      multi-element trains (tests/conftest.py::train_cases) through the reference's numpy sequencer
      propogate_rays_through_optical_system (:1419-1485, sequential branch)  -> pins oracle row f3.

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

    # the BOS generator's own intermediate results (run_simulation_02.py:1328-1551, 999-1056): dot centres and the
    # sunflower point template whose sums are the source coordinates
    extra = {}
    orig_sun, orig_bos = rs.calculate_sunflower_coordinates, rs.generate_bos_lightfield_data

    def sun_wrapped(*a, **k):
        xy = orig_sun(*a, **k)
        extra["tmpl_x"], extra["tmpl_y"] = np.array(xy[0], dtype=np.float64).ravel(), np.array(xy[1], dtype=np.float64).ravel()
        extra["tmpl_args"] = np.array([float(a[0]), float(a[1])])
        return xy

    def bos_wrapped(*a, **k):
        src, gx, gy = orig_bos(*a, **k)
        extra["dot_x"], extra["dot_y"] = np.array(gx, dtype=np.float64).ravel(), np.array(gy, dtype=np.float64).ravel()
        return src, gx, gy

    rs.calculate_sunflower_coordinates, rs.generate_bos_lightfield_data = sun_wrapped, bos_wrapped

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
    rs.calculate_sunflower_coordinates, rs.generate_bos_lightfield_data = orig_sun, orig_bos
    for c in rec.calls:
        c[1].update(extra)
    os.chdir = real_chdir
    os.chdir(OUT)
    shutil.rmtree(work, ignore_errors=True)
    return rec.calls, post, p


def save_case(name, call, post, params, sources_only=False):
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
    if sources_only:
        return
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


def pins_golden():
    """Reference-derived pins for rows whose Python the reference ships: rotation matrices, the
    sunflower template, the numpy ancestor's Mie angle -> irradiance interpolation and pixel area weights."""
    import run_simulation_02 as rs
    import perform_ray_tracing_03 as prt
    prt.long = int
    out = {}
    rng = np.random.default_rng(99)
    # --- calculate_rotation_matrix (run_simulation_02.py:366-392) ---
    ang = np.concatenate([np.zeros((1, 3)), rng.uniform(-0.6, 0.6, (7, 3))])
    out["rot_angles"] = ang
    out["rot_matrices"] = np.stack([np.asarray(rs.calculate_rotation_matrix(*a)) for a in ang])
    # --- calculate_sunflower_coordinates (:999-1056): the per-circle phases come from numpy's global generator ---
    for tag, (dia, npts) in {"a": (600.0, 100.0), "b": (150.0, 25.0), "c": (1000.0, 400.0)}.items():
        np.random.seed(714)
        x, y = rs.calculate_sunflower_coordinates(dia, npts)
        ncirc = int(np.round((dia / 2.0) / np.sqrt(np.pi * (dia / 2.0) ** 2.0 / npts)))
        np.random.seed(714)
        out[f"sun_{tag}_phase"] = np.array([float(np.random.rand(1, 1)[0, 0]) for _ in range(ncirc)])
        out[f"sun_{tag}_args"] = np.array([dia, npts])
        out[f"sun_{tag}_x"], out[f"sun_{tag}_y"] = np.asarray(x, dtype=np.float64).ravel(), np.asarray(y, dtype=np.float64).ravel()
    # --- Mie interpolation inside generate_lightfield_angular_data (perform_ray_tracing_03.py:348-469) ---
    piv = np.load(os.path.join(OUT, "abi_piv.npz"))
    with open(os.path.join(OUT, "abi_piv.json")) as f:
        meta = json.load(f)
    na, nd = meta["scattering"]["num_angles"], meta["scattering"]["num_diameters"]
    angles = piv["scattering_angle"].astype(np.float64)
    table = piv["scattering_irradiance"].astype(np.float64).reshape(na, nd)
    cam_rot = rs.calculate_rotation_matrix(0.05, -0.03, 0.0)
    sd = dict(scattering_angle=angles, scattering_irradiance=table, inverse_rotation_matrix=cam_rot.transpose(),
              beam_propogation_vector=np.matrix([[0.0, 1.0, 0.0]]))
    n_src, L = 24, 64
    src = dict(x=rng.uniform(-7.5e4, 7.5e4, n_src), y=rng.uniform(-7.5e4, 7.5e4, n_src),
               z=823668.3478484906 + rng.uniform(-7.5e3, 7.5e3, n_src), radiance=rng.uniform(50, 500, n_src),
               diameter_index=rng.integers(0, nd, n_src))
    lens_pitch, image_distance = 13125.0, 123529.41176470589
    np.random.seed(123)
    lf = prt.generate_lightfield_angular_data(lens_pitch, image_distance, sd, "mie", src, L, 0, n_src - 1)
    np.random.seed(123)
    u = np.empty((n_src, L)); v = np.empty((n_src, L))
    for n in range(n_src):
        u[n] = np.random.rand(1, L)[0]
        v[n] = np.random.rand(1, L)[0]
    out.update(mie_src_x=src["x"], mie_src_y=src["y"], mie_src_z=src["z"], mie_src_radiance=src["radiance"],
               mie_src_diameter_index=src["diameter_index"].astype(np.int32), mie_u=u, mie_v=v,
               mie_radiance=np.asarray(lf["radiance"]).reshape(n_src, L), mie_theta=np.asarray(lf["theta"]).reshape(n_src, L),
               mie_phi=np.asarray(lf["phi"]).reshape(n_src, L), mie_inverse_rotation=np.asarray(cam_rot.transpose()),
               mie_lens=np.array([lens_pitch, image_distance]))
    # --- intersect_sensor_better (:1488-1595): pixel indices + area weights (MATLAB-style +1.5 offset) ---
    cam = dict(pixel_pitch=17.0, x_pixel_number=1024, y_pixel_number=1024)
    xs = rng.uniform(-8800.0, 8800.0, 4096)
    ys = rng.uniform(-8800.0, 8800.0, 4096)
    ii, jj, w = prt.intersect_sensor_better(cam, xs, ys)
    out.update(sensor_x=xs, sensor_y=ys, sensor_ii=np.asarray(ii, dtype=np.float64), sensor_jj=np.asarray(jj, dtype=np.float64),
               sensor_w=np.asarray(w, dtype=np.float64))
    np.savez_compressed(os.path.join(OUT, "pins.npz"), **out)


def dumps_golden():
    """The ray-dump wire format as the REFERENCE's own reader sees it (light_ray_processing.py:74-207): the oracle writes
    pos_/dir_/intermediate_*.bin for a small BOS pair (tests/conftest.py: dump_pair_calls), the reference's
    load_light_ray_data / load_intermediate_light_ray_data / calculate_lightray_deflections parse them, the parsed arrays
    are the fixture.  Import shims (this script only): the module pulls two plotting helpers of its author's environment
    (not in the repository; neither is used by the readers) and sets a matplotlib option in its pre-3.3 list form."""
    import types
    import matplotlib
    matplotlib.use("Agg")
    matplotlib.rcParams.validate["text.latex.preamble"] = lambda v: "".join(v) if isinstance(v, list) else v
    helper = types.ModuleType("modify_plt_settings")
    helper.modify_plt_settings = lambda plt: plt
    sys.modules.setdefault("modify_plt_settings", helper)
    sys.modules.setdefault("loadmat_functions", types.ModuleType("loadmat_functions"))
    import light_ray_processing as lrp
    numpy_reshape = np.reshape

    def reshape_numpy1(*args, **kw):                    # numpy 1 took reshape(a=..., newshape=...); numpy 2 made `a` positional-only
        if "a" in kw:
            args = (kw.pop("a"),) + args
        if "newshape" in kw:
            kw["shape"] = kw.pop("newshape")
        return numpy_reshape(*args, **kw)
    lrp.np.reshape = reshape_numpy1
    sys.path.insert(0, os.path.join(os.path.dirname(OUT), "..", "oracle"))
    sys.path.insert(0, os.path.dirname(OUT))
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from conftest import DUMP_SLOTS, dump_pair_calls
    from oracle_lib import Oracle
    o = Oracle()
    with tempfile.TemporaryDirectory() as work:
        for call in dump_pair_calls(work):
            o.render(call, interpolation=1)
        pos1, pos2, dir1, dir2 = lrp.load_light_ray_data(work)
        ipos, idir = lrp.load_intermediate_light_ray_data(work, DUMP_SLOTS)
        d_pos, d_dir = lrp.calculate_lightray_deflections(pos1, pos2, dir1, dir2)
    out = {}
    for tag, dct in (("pos1", pos1), ("pos2", pos2), ("dir1", dir1), ("dir2", dir2), ("ipos", ipos), ("idir", idir),
                     ("d_pos", d_pos), ("d_dir", d_dir)):
        for k, v in dct.items():
            out[f"{tag}_{k}"] = np.asarray(v)
    lrp.np.reshape = numpy_reshape
    np.savez_compressed(os.path.join(OUT, "dumps_reference_reader.npz"), **out)
    print("dumps_reference_reader.npz:", {k: v.shape for k, v in out.items() if k.endswith("_x")})


def train_golden():
    """Multi-element trains through the reference's own numpy sequencer, propogate_rays_through_optical_system
    (perform_ray_tracing_03.py:1419-1485).  Its sequential branch (single-member groups, :1438-1451 ->
    propogate_rays_through_single_element) runs in this image; the simultaneous-elements branch (:1453-1472) does not.
    Scenes: tests/conftest.py::train_cases.  The oracle generates each scene's rays in f32 exactly as a render would
    (source-major, lens sample k uses the srand(10) table entry k); the reference propagates them in float64; stored per
    case: the rays in, the reference's rays out (NaN = destroyed), radiance out, and where the straight line of each
    surviving ray meets the sensor plane z = z_sensor (plain geometry on the reference's output, so that the GPU's
    sensor-position dump can be compared through the C-ABI)."""
    import run_simulation_02  # noqa: F401  (resolves the circular import)
    import perform_ray_tracing_03 as prt
    prt.long = int
    sys.path.insert(0, os.path.join(os.path.dirname(OUT), "..", "oracle"))
    sys.path.insert(0, os.path.dirname(OUT))
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from conftest import train_cases
    from oracle_lib import Oracle
    o = Oracle()
    names = {"l": "lens", "a": "aperture"}
    out = {}
    for case, call in train_cases().items():
        L = int(call.lightray_number_per_particle)
        r1, r2 = o.rand_table(L)
        pos, direction, rad = [], [], []
        for s in range(call.num_sources):
            p, d, r = o.generate_rays(call, s, r1, r2)
            pos.append(p); direction.append(d); rad.append(r)
        pos, direction, rad = np.concatenate(pos), np.concatenate(direction), np.concatenate(rad)
        elements = []
        for e in call.elements:
            g, pr = e["element_geometry"], e["element_properties"]
            elements.append(dict(element_type=names[e["element_type"]],
                                 element_geometry=dict(pitch=g["pitch"], vertex_distance=g["vertex_distance"],
                                                       front_surface_radius=g["front_surface_radius"],
                                                       back_surface_radius=g["back_surface_radius"]),
                                 element_properties=dict(refractive_index=pr["refractive_index"], abbe_number=np.nan,
                                                         transmission_ratio=pr.get("transmission_ratio", 1.0),
                                                         absorbance_rate=pr.get("absorbance_rate", 0.0))))
        lrd = dict(ray_source_coordinates=pos.astype(np.float64), ray_propogation_direction=direction.astype(np.float64),
                   ray_wavelength=np.full(len(pos), float(call.beam_wavelength)), ray_radiance=rad.copy())
        res = prt.propogate_rays_through_optical_system(elements, np.array(call.element_center, np.float64),
                                                        np.array(call.element_plane_parameters, np.float64),
                                                        np.array(call.element_system_index), lrd)
        po = np.asarray(res["ray_source_coordinates"], np.float64)
        do = np.asarray(res["ray_propogation_direction"], np.float64)
        with np.errstate(invalid="ignore", divide="ignore"):
            t = (float(call.camera["z_sensor"]) - po[:, 2]) / do[:, 2]
            hit = po[:, :2] + do[:, :2] * t[:, None]
        out[f"{case}_in"] = np.concatenate([pos, direction], 1)                     # f32: the oracle's generated rays
        out[f"{case}_in_radiance"] = rad
        out[f"{case}_out"] = np.concatenate([po, do], 1)
        out[f"{case}_out_radiance"] = np.asarray(res["ray_radiance"], np.float64)
        out[f"{case}_sensor_xy"] = hit
        print(f"train {case}: {len(pos)} rays, {np.isnan(po[:, 0]).mean():.3f} destroyed")
    np.savez_compressed(os.path.join(OUT, "train_f64.npz"), **out)


def main():
    sys.path.insert(0, os.path.join(REF, "python_codes"))
    if "--dumps-only" in sys.argv:
        dumps_golden()
        return
    if "--train-only" in sys.argv:
        train_golden()
        return

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
    pins_golden()
    dumps_golden()
    train_golden()
    # the sample cases at their REAL size (50 000 particles x 10 000 rays; the full dot grid x 500 rays): source arrays
    # + scalars only
    calls, post, params = run_case("piv", lambda p: p["particle_field"].__setitem__("frame_vector", np.array([1])))
    save_case("piv_full", calls[0], None, params, sources_only=True)
    calls, post, params = run_case("bos", lambda p: None)
    save_case("bos_full_im1", calls[0], None, params, sources_only=True)
    save_case("bos_full_im2", calls[1], None, params, sources_only=True)
    # the one sample volume the reference ships (data, 1 MiB)
    shutil.copyfile(os.path.join(REF, "sample-data", "bos", "sample-density.nrrd"),
                    os.path.join(OUT, "sample-density.nrrd"))
    os.chmod(os.path.join(OUT, "sample-density.nrrd"), 0o644)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
