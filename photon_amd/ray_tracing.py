"""Host-side mirror of photon's marshalling layer for the ray-tracing hot path.

This is the repo's own counterpart of the reference's
``python_codes/perform_ray_tracing_03.py:1631-2291`` (``prepare_data_for_cytpes_call`` and
``perform_ray_tracing_03``): it packs an experiment description into the C structs of
``include/parallel_ray_tracing.h`` and calls ``start_ray_tracing`` through ctypes, then applies
the reference's sensor post-processing.  photon itself keeps using its own file unchanged
(see INTEGRATION.md); this module exists so that the tests, ``bench.py`` and ``smoke()`` drive
the library through the identical C-ABI, and it is pinned against fixtures captured from the
reference's marshalling code (``tests/golden/abi_*.npz``).

Nothing here computes rays: without the HIP library the calls fail loudly.
"""
from __future__ import annotations

import ctypes
import json
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

# ----------------------------------------------------------------------------------------------
# ctypes mirrors of the wire structs (include/parallel_ray_tracing.h section 1;
# reference: perform_ray_tracing_03.py:1651-1659, 1708-1720, 1751-1786, 1838-1852)
# ----------------------------------------------------------------------------------------------


class scattering_data_struct(ctypes.Structure):
    _fields_ = [
        ("inverse_rotation_matrix", ctypes.c_float * 9),
        ("beam_propagation_vector", ctypes.c_float * 3),
        ("scattering_angle", ctypes.c_void_p),
        ("scattering_irradiance", ctypes.c_void_p),
        ("num_angles", ctypes.c_int),
        ("num_diameters", ctypes.c_int),
    ]


class lightfield_source_struct(ctypes.Structure):
    _fields_ = [
        ("lightray_number_per_particle", ctypes.c_int),
        ("source_point_number", ctypes.c_int),
        ("diameter_index", ctypes.c_void_p),
        ("radiance", ctypes.c_void_p),
        ("x", ctypes.c_void_p),
        ("y", ctypes.c_void_p),
        ("z", ctypes.c_void_p),
        ("num_particles", ctypes.c_int),
        ("z_offset", ctypes.c_float),
        ("object_distance", ctypes.c_float),
    ]


class element_geometry_struct(ctypes.Structure):
    _fields_ = [
        ("front_surface_radius", ctypes.c_float),
        ("front_surface_spherical", ctypes.c_bool),
        ("back_surface_radius", ctypes.c_float),
        ("back_surface_spherical", ctypes.c_bool),
        ("pitch", ctypes.c_float),
        ("vertex_distance", ctypes.c_double),
    ]


class element_properties_struct(ctypes.Structure):
    _fields_ = [
        ("abbe_number", ctypes.c_float),
        ("absorbance_rate", ctypes.c_float),
        ("refractive_index", ctypes.c_double),
        ("thin_lens_focal_length", ctypes.c_float),
        ("transmission_ratio", ctypes.c_float),
    ]


class element_data_struct(ctypes.Structure):
    _fields_ = [
        ("axial_offset_distances", ctypes.c_double * 2),
        ("element_geometry", element_geometry_struct),
        ("element_number", ctypes.c_float),
        ("element_properties", element_properties_struct),
        ("element_type", ctypes.c_char),
        ("elements_coplanar", ctypes.c_float),
        ("rotation_angles", ctypes.c_double * 3),
        ("z_inter_element_distance", ctypes.c_float),
    ]


class camera_design_struct(ctypes.Structure):
    _fields_ = [
        ("pixel_bit_depth", ctypes.c_int),
        ("pixel_gain", ctypes.c_float),
        ("pixel_pitch", ctypes.c_float),
        ("x_camera_angle", ctypes.c_float),
        ("y_camera_angle", ctypes.c_float),
        ("x_pixel_number", ctypes.c_int),
        ("y_pixel_number", ctypes.c_int),
        ("z_sensor", ctypes.c_float),
        ("diffraction_diameter", ctypes.c_float),
        ("implement_diffraction", ctypes.c_bool),
        ("rotation_matrix", ctypes.c_float * 9),
        ("inverse_rotation_matrix", ctypes.c_float * 9),
    ]


START_RAY_TRACING_ARGTYPES = [
    ctypes.c_float, ctypes.c_float, ctypes.POINTER(scattering_data_struct), ctypes.c_char_p,
    ctypes.POINTER(lightfield_source_struct), ctypes.c_int, ctypes.c_float, ctypes.c_float,
    ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(element_data_struct), ctypes.c_void_p,
    ctypes.c_void_p, ctypes.POINTER(camera_design_struct), ctypes.c_void_p,
    ctypes.c_bool, ctypes.c_char_p, ctypes.c_bool, ctypes.c_char_p, ctypes.c_char_p,
    ctypes.c_int, ctypes.c_int, ctypes.c_bool, ctypes.c_float, ctypes.c_bool, ctypes.c_float,
    ctypes.c_float, ctypes.c_bool, ctypes.c_int,
]

LENS_MODEL_TO_ELEMENT_TYPE = {"thin-lens": b"t", "apparent": b"n"}   # anything else -> b'l' (:1803-1808)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


@dataclass
class RayTracingCall:
    """All arguments of one ``start_ray_tracing`` call, as plain Python / numpy values."""

    lens_pitch: float
    image_distance: float
    scattering_type: str                      # 'mie' | 'diffuse'
    src_x: np.ndarray
    src_y: np.ndarray
    src_z: np.ndarray
    src_radiance: np.ndarray                  # float64
    src_diameter_index: np.ndarray            # int32
    lightray_number_per_particle: int
    source_point_number: int
    z_offset: float
    object_distance: float
    beam_wavelength: float
    aperture_f_number: float
    elements: list                            # list of dicts (see element_from_dict)
    element_center: np.ndarray                # float64 [n,3]
    element_plane_parameters: np.ndarray      # float64 [n,4]
    element_system_index: np.ndarray          # int32 [n]
    camera: dict
    simulate_density_gradients: bool = False
    density_grad_filename: str = ""
    ray_tracing_algorithm: int = 0
    ray_cone_pitch_ratio: float = 1e-4
    save_lightrays: bool = False
    lightray_position_save_path: str = ""
    lightray_direction_save_path: str = ""
    num_lightrays_save: int = 0
    add_pos_noise: bool = False
    pos_noise_std: float = 0.0
    add_ngrad_noise: bool = False
    ngrad_noise_std: float = 0.0
    save_intermediate_ray_data: bool = False
    num_intermediate_positions_save: int = 0
    scattering: Optional[dict] = None         # mie only: inverse_rotation_matrix[9], beam[3]
    scattering_angle: Optional[np.ndarray] = None
    scattering_irradiance: Optional[np.ndarray] = None   # [num_angles, num_diameters]
    _keep: list = field(default_factory=list, repr=False)

    # ---- sizes -------------------------------------------------------------------------
    @property
    def num_sources(self) -> int:
        return int(self.src_x.size)

    @property
    def num_rays(self) -> int:
        return self.num_sources * int(self.lightray_number_per_particle)

    @property
    def image_shape(self):
        return int(self.camera["y_pixel_number"]), int(self.camera["x_pixel_number"])

    def new_image(self) -> np.ndarray:
        return np.zeros(self.image_shape, dtype=np.float32)

    # ---- struct packing (what perform_ray_tracing_03.py:1651-1878 does) -------------------
    def pack(self):
        keep = self._keep
        keep.clear()
        sd = scattering_data_struct()
        if self.scattering_type == "mie":
            ang = _f32(self.scattering_angle)
            irr = _f32(self.scattering_irradiance)
            keep += [ang, irr]
            sd.scattering_angle = ang.ctypes.data
            sd.scattering_irradiance = irr.ctypes.data
            sd.num_angles = int(irr.shape[0])
            sd.num_diameters = int(irr.shape[1])
            for i in range(3):
                sd.beam_propagation_vector[i] = float(self.scattering["beam_propagation_vector"][i])
            for i in range(9):
                sd.inverse_rotation_matrix[i] = float(self.scattering["inverse_rotation_matrix"][i])
        else:                                   # :1685-1701 NULL tables, NaN vectors
            sd.scattering_angle = None
            sd.scattering_irradiance = None
            sd.num_angles = 0
            sd.num_diameters = 0
            for i in range(3):
                sd.beam_propagation_vector[i] = float("nan")
            for i in range(9):
                sd.inverse_rotation_matrix[i] = float("nan")

        x, y, z = _f32(self.src_x), _f32(self.src_y), _f32(self.src_z)
        rad = np.ascontiguousarray(np.asarray(self.src_radiance, dtype=np.float64))
        dia = np.ascontiguousarray(np.asarray(self.src_diameter_index, dtype=np.int32))
        keep += [x, y, z, rad, dia]
        ls = lightfield_source_struct()
        ls.lightray_number_per_particle = int(self.lightray_number_per_particle)
        ls.source_point_number = int(self.source_point_number)
        ls.diameter_index = dia.ctypes.data
        ls.radiance = rad.ctypes.data
        ls.x, ls.y, ls.z = x.ctypes.data, y.ctypes.data, z.ctypes.data
        ls.num_particles = int(dia.size)
        ls.z_offset = float(self.z_offset)
        ls.object_distance = float(self.object_distance)

        n = len(self.elements)
        elems = (element_data_struct * n)()
        for i, e in enumerate(self.elements):
            element_from_dict(e, elems[i])
        centers = np.ascontiguousarray(np.asarray(self.element_center, dtype=np.float64).reshape(n, 3))
        planes = np.ascontiguousarray(np.asarray(self.element_plane_parameters, dtype=np.float64).reshape(n, 4))
        sysidx = np.ascontiguousarray(np.asarray(self.element_system_index, dtype=np.int32).reshape(n))
        keep += [centers, planes, sysidx]

        cam = camera_from_dict(self.camera)
        return sd, ls, elems, centers, planes, sysidx, cam

    def invoke(self, fn, image: np.ndarray, extra=()):
        """Call ``fn`` (start_ray_tracing or an ABI-compatible function) on ``image`` in place.

        ``image`` must be C-contiguous float32 of shape ``image_shape``; it is accumulated
        into, exactly as the reference does (parallel_ray_tracing.cu:3309,3675)."""
        assert image.dtype == np.float32 and image.flags.c_contiguous
        assert image.shape == self.image_shape
        sd, ls, elems, centers, planes, sysidx, cam = self.pack()
        fn(ctypes.c_float(self.lens_pitch), ctypes.c_float(self.image_distance), ctypes.byref(sd),
           self.scattering_type.encode("utf-8"), ctypes.byref(ls),
           int(self.lightray_number_per_particle), ctypes.c_float(self.beam_wavelength),
           ctypes.c_float(self.aperture_f_number), len(self.elements), centers.ctypes.data, elems,
           planes.ctypes.data, sysidx.ctypes.data, ctypes.byref(cam), image.ctypes.data,
           bool(self.simulate_density_gradients), self.density_grad_filename.encode("utf-8"),
           bool(self.save_lightrays), self.lightray_position_save_path.encode("utf-8"),
           self.lightray_direction_save_path.encode("utf-8"), int(self.num_lightrays_save),
           int(self.ray_tracing_algorithm), bool(self.add_pos_noise), ctypes.c_float(self.pos_noise_std),
           bool(self.add_ngrad_noise), ctypes.c_float(self.ngrad_noise_std),
           ctypes.c_float(self.ray_cone_pitch_ratio), bool(self.save_intermediate_ray_data),
           int(self.num_intermediate_positions_save), *extra)
        return image

    # ---- fixtures -------------------------------------------------------------------------
    @classmethod
    def from_fixture(cls, json_path: str, npz_path: str, density_dir: Optional[str] = None):
        """Rebuild a call from tests/golden/abi_<case>.{json,npz} (captured from the reference)."""
        with open(json_path) as f:
            j = json.load(f)
        a = np.load(npz_path)
        s = j["scalars"]
        src = j["source"]
        grad = s["density_grad_filename"]
        if grad and density_dir is not None:
            grad = os.path.join(density_dir, grad)
        call = cls(
            lens_pitch=s["lens_pitch"], image_distance=s["image_distance"],
            scattering_type=s["scattering_type"],
            src_x=a["src_x"], src_y=a["src_y"], src_z=a["src_z"], src_radiance=a["src_radiance"],
            src_diameter_index=a["src_diameter_index"],
            lightray_number_per_particle=s["lightray_number_per_particle"],
            source_point_number=src["source_point_number"], z_offset=src["z_offset"],
            object_distance=src["object_distance"], beam_wavelength=s["beam_wavelength"],
            aperture_f_number=s["aperture_f_number"], elements=j["elements"],
            element_center=a["element_center"], element_plane_parameters=a["element_plane_parameters"],
            element_system_index=a["element_system_index"], camera=j["camera"],
            simulate_density_gradients=s["simulate_density_gradients"], density_grad_filename=grad,
            ray_tracing_algorithm=s["ray_tracing_algorithm"],
            ray_cone_pitch_ratio=s["ray_cone_pitch_ratio"], save_lightrays=s["save_lightrays"],
            num_lightrays_save=s["num_lightrays_save"])
        if s["scattering_type"] == "mie":
            sc = j["scattering"]
            call.scattering = dict(inverse_rotation_matrix=sc["inverse_rotation_matrix"],
                                   beam_propagation_vector=sc["beam_propogation_vector"])
            call.scattering_angle = a["scattering_angle"]
            call.scattering_irradiance = a["scattering_irradiance"].reshape(sc["num_angles"], sc["num_diameters"])
        return call


def element_from_dict(e: dict, out: Optional[element_data_struct] = None) -> element_data_struct:
    """Fill an element_data_t from the nested dict layout photon uses
    (perform_ray_tracing_03.py:1796-1832)."""
    s = out if out is not None else element_data_struct()
    for j in range(2):
        s.axial_offset_distances[j] = float(e.get("axial_offset_distances", (0.0, 0.0))[j])
    g, p = e["element_geometry"], e["element_properties"]
    s.element_geometry.front_surface_radius = float(g["front_surface_radius"])
    s.element_geometry.front_surface_spherical = bool(g.get("front_surface_spherical", True))
    s.element_geometry.back_surface_radius = float(g["back_surface_radius"])
    s.element_geometry.back_surface_spherical = bool(g.get("back_surface_spherical", True))
    s.element_geometry.pitch = float(g["pitch"])
    s.element_geometry.vertex_distance = float(g["vertex_distance"])
    s.element_number = float(e.get("element_number", float("nan")))
    s.element_properties.abbe_number = float(p.get("abbe_number", float("nan")))
    s.element_properties.absorbance_rate = float(p.get("absorbance_rate", 0.0))
    s.element_properties.refractive_index = float(p["refractive_index"])
    s.element_properties.thin_lens_focal_length = float(p["thin_lens_focal_length"])
    s.element_properties.transmission_ratio = float(p.get("transmission_ratio", 1.0))
    t = e.get("element_type", "l")
    s.element_type = t.encode("utf-8") if isinstance(t, str) else t
    s.elements_coplanar = float(e.get("elements_coplanar", float("nan")))
    for j in range(3):
        s.rotation_angles[j] = float(e.get("rotation_angles", (0.0, 0.0, 0.0))[j])
    s.z_inter_element_distance = float(e.get("z_inter_element_distance", 0.0))
    return s


def camera_from_dict(c: dict) -> camera_design_struct:
    """perform_ray_tracing_03.py:1855-1874."""
    s = camera_design_struct()
    s.pixel_bit_depth = int(c["pixel_bit_depth"])
    s.pixel_gain = float(c["pixel_gain"])
    s.pixel_pitch = float(c["pixel_pitch"])
    s.x_camera_angle = float(c.get("x_camera_angle", 0.0))
    s.y_camera_angle = float(c.get("y_camera_angle", 0.0))
    s.x_pixel_number = int(c["x_pixel_number"])
    s.y_pixel_number = int(c["y_pixel_number"])
    s.z_sensor = float(c.get("z_sensor", 0.0))
    s.diffraction_diameter = float(c.get("diffraction_diameter", 0.0))
    s.implement_diffraction = bool(c.get("implement_diffraction", False))
    rot = np.asarray(c.get("rotation_matrix", np.eye(3)), dtype=np.float32).reshape(9)
    inv = np.asarray(c.get("inverse_rotation_matrix", np.eye(3)), dtype=np.float32).reshape(9)
    for i in range(9):
        s.rotation_matrix[i] = float(rot[i])
        s.inverse_rotation_matrix[i] = float(inv[i])
    return s


# ----------------------------------------------------------------------------------------------
# Single-lens camera geometry: the numbers perform_ray_tracing_03.py:2016-2078 derives before the
# call (image distance, principal plane, lens position) for photon's one-lens optical system
# (run_simulation_02.py:259-363 solves thickness and refractive index).
# ----------------------------------------------------------------------------------------------


def calculate_rotation_matrix(theta_x: float, theta_y: float, theta_z: float = 0.0) -> np.ndarray:
    """World -> camera rotation of photon's driver (run_simulation_02.py:366-392): R = Rx . Ry . Rz with the
    reference's sign convention; camera_design's `rotation_matrix` is this for (x_camera_angle, y_camera_angle, 0)
    and `inverse_rotation_matrix` its transpose (:1748-1753).  Pinned against the reference's outputs
    (tests/golden/pins.npz)."""
    cx, sx, cy, sy, cz, sz = (np.cos(theta_x), np.sin(theta_x), np.cos(theta_y), np.sin(theta_y), np.cos(theta_z),
                              np.sin(theta_z))
    rx = np.array([[1.0, 0.0, 0.0], [0.0, cx, sx], [0.0, -sx, cx]])
    ry = np.array([[cy, 0.0, -sy], [0.0, 1.0, 0.0], [sy, 0.0, cy]])
    rz = np.array([[cz, sz, 0.0], [-sz, cz, 0.0], [0.0, 0.0, 1.0]])
    return rx @ ry @ rz


def single_lens_camera(focal_length: float, aperture_f_number: float, object_distance: float,
                       lens_radius_of_curvature: float, lens_model: str = "general") -> dict:
    f, R = float(focal_length), float(lens_radius_of_curvature)
    pitch = f / float(aperture_f_number)
    thickness = 0.0 if lens_model == "thin-lens" else 2.0 * (R - np.sqrt(R ** 2 - (pitch / 2.0) ** 2))
    disc = np.sqrt(-4.0 * thickness * f + (2.0 * f + R) ** 2)
    den = 2.0 * f * (thickness - 2.0 * R)
    cands = [(2.0 * thickness * f - 2.0 * f * R - R ** 2 - R * disc) / den,
             (2.0 * thickness * f - 2.0 * f * R - R ** 2 + R * disc) / den]
    n = min(c for c in cands if np.isreal(c) and c >= 1.0)
    image_distance = 1.0 / (1.0 / f - 1.0 / object_distance)
    h2 = -(f * (n - 1.0) * thickness) / (R * n)
    v2 = image_distance + h2
    v1 = v2 + thickness
    z_lens = (v1 + v2) / 2.0
    # object-side geometry (run_simulation_02.py:866-879): back radius is -R
    h1 = -(f * (n - 1.0) * thickness) / (-R * n)
    z_object = v1 - h1 + object_distance
    z_offset = z_object - object_distance
    element = dict(
        element_type=LENS_MODEL_TO_ELEMENT_TYPE.get(lens_model, b"l").decode(),
        element_geometry=dict(front_surface_radius=+R, back_surface_radius=-R, pitch=pitch,
                              vertex_distance=thickness),
        element_properties=dict(refractive_index=float(n), thin_lens_focal_length=f,
                                abbe_number=float("nan"), absorbance_rate=0.0, transmission_ratio=1.0))
    return dict(lens_pitch=pitch, image_distance=image_distance, z_lens=z_lens, element=element,
                element_center=np.array([[0.0, 0.0, z_lens]]),
                element_plane_parameters=np.array([[0.0, 0.0, 1.0, -z_lens]]),
                element_system_index=np.array([1], dtype=np.int32),
                refractive_index=float(n), thickness=float(thickness), h2_principal_plane=h2,
                v1_vertex_plane=v1, v2_vertex_plane=v2, z_object=z_object, z_offset=z_offset,
                object_distance=float(object_distance))


# ----------------------------------------------------------------------------------------------
# Sensor post-processing (perform_ray_tracing_03.py:2190-2247): the step right after the hot path.
# ----------------------------------------------------------------------------------------------


def crop_window(n_rows: int, n_cols: int, r_crop: int, c_crop: int):
    """The reference's centre crop (perform_ray_tracing_03.py:2250-2259): rows [nr/2 - r/2, nr/2 + r/2 - 1) with
    integer division -- one row and one column FEWER than asked for, as its slice has it."""
    r0, c0 = n_rows // 2 - r_crop // 2, n_cols // 2 - c_crop // 2
    return slice(r0, n_rows // 2 + r_crop // 2 - 1), slice(c0, n_cols // 2 + c_crop // 2 - 1)


def postprocess_image(I_raw: np.ndarray, pixel_gain: float, pixel_bit_depth: int,
                      intensity_rescaling: bool = True, image_noise: float = 0.0,
                      rng: Optional[np.random.Generator] = None) -> np.ndarray:
    """Raw float image -> uint16 sensor image: (noise) -> clip<0 -> 10^(gain/20) ->
    normalise to the brightest pixel -> round to bit depth -> stretch to 16 bit.  Host (numpy) form, pinned pixel
    for pixel against the reference's output; the device form is photon_postprocess_u16
    (PhotonLibrary.postprocess_u16), pinned against the same fixtures."""
    I = np.array(I_raw, dtype=np.float32, copy=True)
    if image_noise > 0.0:
        rng = rng or np.random.default_rng()
        I = I + rng.normal(0.0, image_noise * 100.0, size=I.shape).astype(I.dtype)
    I[I < 0.0] = 0.0
    if intensity_rescaling:
        I[~np.isfinite(I)] = 0.0
        I *= 10 ** (pixel_gain / 20.0)
        bits = int(pixel_bit_depth)
        if np.max(I) > 0.0:
            I = (2 ** bits - 1) * I / np.max(I)
        I = np.round(I)
        I *= (2 ** 16 - 1.0) / (2 ** bits - 1.0)
    return np.uint16(I)


def write_tiff_u16(path: str, image: np.ndarray) -> str:
    """Minimal baseline TIFF writer (little-endian, one strip, 16-bit grayscale, uncompressed) for the
    uint16 sensor image -- what the reference writes with its vendored tifffile
    (run_simulation_02.py:1864, 2052, 2086).  Readable by any TIFF reader."""
    import struct
    img = np.ascontiguousarray(image, dtype="<u2")
    h, w = img.shape
    data = img.tobytes()
    entries = [  # (tag, type, count, value)   type 3 = SHORT, 4 = LONG
        (256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, 16), (259, 3, 1, 1), (262, 3, 1, 1),
        (273, 4, 1, 8), (277, 3, 1, 1), (278, 4, 1, h), (279, 4, 1, len(data)), (339, 3, 1, 1),
    ]
    ifd_offset = 8 + len(data) + (len(data) & 1)
    with open(path, "wb") as f:
        f.write(struct.pack("<2sHI", b"II", 42, ifd_offset))
        f.write(data)
        if len(data) & 1:
            f.write(b"\0")
        f.write(struct.pack("<H", len(entries)))
        for tag, typ, count, value in entries:
            f.write(struct.pack("<HHI", tag, typ, count))
            f.write(struct.pack("<HH", value, 0) if typ == 3 else struct.pack("<I", value))
        f.write(struct.pack("<I", 0))
    return path


def read_tiff_u16(path: str) -> np.ndarray:
    """Reader for the files write_tiff_u16 produces (tests)."""
    import struct
    with open(path, "rb") as f:
        buf = f.read()
    assert buf[:4] == b"II*\0"
    (ifd,) = struct.unpack_from("<I", buf, 4)
    (n,) = struct.unpack_from("<H", buf, ifd)
    tags = {}
    for i in range(n):
        tag, typ, count = struct.unpack_from("<HHI", buf, ifd + 2 + 12 * i)
        tags[tag] = struct.unpack_from("<H" if typ == 3 else "<I", buf, ifd + 2 + 12 * i + 8)[0]
    w, h, off = tags[256], tags[257], tags[273]
    return np.frombuffer(buf, dtype="<u2", count=w * h, offset=off).reshape(h, w).copy()


def save_images(image_raw: np.ndarray, image_u16: np.ndarray, tif_path: str, raw_path: str):
    """What the driver does with the two results of a render (run_simulation_02.py:2049-2058):
    uint16 TIFF + raw float32 dump."""
    write_tiff_u16(tif_path, image_u16)
    np.asarray(image_raw, dtype=np.float32).tofile(raw_path)


# ----------------------------------------------------------------------------------------------
# Library loading
# ----------------------------------------------------------------------------------------------


def bind_start_ray_tracing(lib: ctypes.CDLL, name: str = "start_ray_tracing", extra_argtypes=()):
    fn = getattr(lib, name)
    fn.argtypes = START_RAY_TRACING_ARGTYPES + list(extra_argtypes)
    fn.restype = None
    return fn
