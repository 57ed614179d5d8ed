"""ctypes access to the CPU parity oracle (oracle/libphoton_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under photon_amd/ imports this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # repo root (this file: oracle/)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from photon_amd.library import photon_volume_info_t  # noqa: E402  (struct layout only)
from photon_amd.ray_tracing import RayTracingCall, bind_start_ray_tracing, element_from_dict  # noqa: E402

ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libphoton_oracle.so")


class oracle_stats_t(ctypes.Structure):
    _fields_ = [("rays_launched", ctypes.c_uint64), ("rays_on_sensor", ctypes.c_uint64),
                ("rk_iterations", ctypes.c_uint64), ("volume_samples", ctypes.c_uint64),
                ("sensor_taps", ctypes.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def build_oracle(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "photon_oracle.cpp")
    stale = (not os.path.exists(ORACLE_SO)) or any(
        os.path.getmtime(p) > os.path.getmtime(ORACLE_SO)
        for p in (src, os.path.join(ROOT, "include", "parallel_ray_tracing.h"),
                  os.path.join(ROOT, "include", "photon_det_math.h")))
    if force or stale:
        subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True, stdout=sys.stderr)
    return ORACLE_SO


ORACLE_FMA_SO = os.path.join(ORACLE_DIR, "libphoton_oracle_fma.so")


def build_oracle_fma() -> str:
    """The contracted-FMA build of the same source (oracle/Makefile, target `fma`): a sensitivity probe, never the reference."""
    subprocess.run(["make", "-C", ORACLE_DIR, "-s", "fma"], check=True, stdout=sys.stderr)
    return ORACLE_FMA_SO


ORACLE_LITERAL_SO = os.path.join(ORACLE_DIR, "libphoton_oracle_literal.so")


def build_oracle_literal() -> str:
    """The build with the reference's literal B-spline weight expressions (oracle/Makefile, target `literal`): a sensitivity
    probe for the one place where oracle and product deviate from the reference's spelling together; never the reference."""
    subprocess.run(["make", "-C", ORACLE_DIR, "-s", "literal"], check=True, stdout=sys.stderr)
    return ORACLE_LITERAL_SO


ORACLE_LIBM_ERF_SO = os.path.join(ORACLE_DIR, "libphoton_oracle_libm_erf.so")


def build_oracle_libm_erf() -> str:
    """The build whose erf splat calls glibc's erf() instead of photon_det_erf (oracle/Makefile, target `libm_erf`): a
    sensitivity probe; never the reference."""
    subprocess.run(["make", "-C", ORACLE_DIR, "-s", "libm_erf"], check=True, stdout=sys.stderr)
    return ORACLE_LIBM_ERF_SO


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Oracle:
    def __init__(self, contracted: bool = False, literal_bspline: bool = False, libm_erf: bool = False):
        assert contracted + literal_bspline + libm_erf <= 1
        self.lib = ctypes.CDLL(build_oracle_fma() if contracted else build_oracle_literal() if literal_bspline
                               else build_oracle_libm_erf() if libm_erf else build_oracle())
        L = self.lib
        self._start = bind_start_ray_tracing(L, "oracle_start_ray_tracing",
                                             [ctypes.c_int, ctypes.c_int, ctypes.POINTER(oracle_stats_t)])
        self._render_vol = bind_start_ray_tracing(L, "oracle_render_with_volume",
                                                  [ctypes.c_void_p, ctypes.POINTER(oracle_stats_t)])
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_volume_from_density.restype = ctypes.c_void_p
        L.oracle_volume_from_density.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.oracle_volume_load_nrrd.restype = ctypes.c_void_p
        L.oracle_volume_load_nrrd.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
        L.oracle_volume_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(photon_volume_info_t)]
        L.oracle_volume_download.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.oracle_volume_sample.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_trace_volume_rays.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_volume_free.argtypes = [ctypes.c_void_p]
        L.oracle_det_eval.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_ray_sphere_intersection.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p,
                                                     ctypes.c_void_p, ctypes.c_char, ctypes.c_void_p]
        L.oracle_axis_distance.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4
        L.oracle_single_element.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]

    # ---- full pipeline --------------------------------------------------------------------
    def render(self, call: RayTracingCall, image=None, interpolation: int = 1, tex_frac_bits: int = 8):
        if image is None:
            image = call.new_image()
        st = oracle_stats_t()
        call.invoke(self._start, image, extra=(int(interpolation), int(tex_frac_bits), ctypes.byref(st)))
        return image, st

    def render_with_volume(self, call: RayTracingCall, volume, image=None):
        """Ray loop only, on a prebuilt OracleVolume (or None)."""
        if image is None:
            image = call.new_image()
        st = oracle_stats_t()
        call.invoke(self._render_vol, image, extra=(volume.handle if volume is not None else None, ctypes.byref(st)))
        return image, st

    def num_threads(self) -> int:
        return int(self.lib.oracle_num_threads())

    def set_noise_seed(self, seed: int):
        self.lib.oracle_set_noise_seed(ctypes.c_uint64(int(seed)))

    def set_element_train(self, mode: int):
        self.lib.oracle_set_element_train(int(mode))

    def normal2(self, seed: int, n: int, draw: int = 0, stream: int = 1):
        out = np.empty((n, 2), np.float32)
        self.lib.oracle_normal2(ctypes.c_uint64(int(seed)), int(n), ctypes.c_uint32(draw), ctypes.c_uint32(stream), _p(out))
        return out

    def set_num_threads(self, n: int):
        self.lib.oracle_set_num_threads(int(n))

    def rand_table(self, n):
        r1, r2 = np.empty(n, np.float32), np.empty(n, np.float32)
        self.lib.oracle_rand_table(n, _p(r1), _p(r2))
        return r1, r2

    def det_erf(self, x):
        """photon_det_erf (include/photon_det_math.h) as this build evaluates it."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        self.lib.oracle_det_erf.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self.lib.oracle_det_erf(int(x.size), _p(x), _p(out))
        return out

    def det_div_rcp_mismatches(self, a, b: float) -> int:
        """How many of a[i] / b differ from photon_det_div_rcp(a[i], b, 1 / b)."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        f = self.lib.oracle_det_div_rcp_mismatches
        f.restype = ctypes.c_longlong
        f.argtypes = [ctypes.c_longlong, ctypes.c_void_p, ctypes.c_double]
        return int(f(int(a.size), _p(a), float(b)))

    def det_eval(self, fn: int, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self.lib.oracle_det_eval(fn, x.size, _p(x), _p(y))
        return y

    # ---- volume ---------------------------------------------------------------------------
    def volume_from_density(self, rho, spacing, origin, interpolation=1, tex_frac_bits=8):
        rho = np.ascontiguousarray(rho, dtype=np.float32)
        nz, ny, nx = rho.shape
        sp = np.ascontiguousarray(spacing, dtype=np.float64)
        og = np.ascontiguousarray(origin, dtype=np.float64)
        h = self.lib.oracle_volume_from_density(_p(rho), nx, ny, nz, _p(sp), _p(og), interpolation, tex_frac_bits)
        return OracleVolume(self, h)

    def volume_gaussian(self, n, spacing, origin, rho0, amp, centre, sigma, interpolation=1, tex_frac_bits=8):
        nx, ny, nz = (n, n, n) if np.isscalar(n) else n
        sp = np.ascontiguousarray(np.broadcast_to(np.asarray(spacing, np.float64), (3,)))
        og = np.ascontiguousarray(origin, dtype=np.float64)
        c = np.ascontiguousarray(centre, dtype=np.float64)
        self.lib.oracle_volume_gaussian.restype = ctypes.c_void_p
        h = self.lib.oracle_volume_gaussian(int(nx), int(ny), int(nz), _p(sp), _p(og), ctypes.c_double(rho0),
                                            ctypes.c_double(amp), _p(c), ctypes.c_double(sigma), int(interpolation),
                                            int(tex_frac_bits))
        return OracleVolume(self, h)

    # ---- single device functions, for the reference-derived pins ------------------------------
    def generate_rays(self, call: RayTracingCall, source: int, r1, r2):
        """Rays of one source for explicit lens samples (r1, r2): pos/dir [n][3] f32, radiance [n] f64
        (generate_lightfield_angular_data, parallel_ray_tracing.cu:71-237)."""
        sd, ls, elems, centers, planes, sysidx, cam = call.pack()
        r1 = np.ascontiguousarray(r1, dtype=np.float32)
        r2 = np.ascontiguousarray(r2, dtype=np.float32)
        n = r1.size
        pos, direction, rad = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32), np.empty(n, np.float64)
        f = self.lib.oracle_generate_rays
        f.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                      ctypes.c_float, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        f(call.lens_pitch, call.image_distance, ctypes.addressof(sd), 1 if call.scattering_type == "mie" else 0,
          float(np.float32(call.src_x[source])), float(np.float32(call.src_y[source])), float(np.float32(call.src_z[source])),
          float(call.src_radiance[source]), int(call.src_diameter_index[source]), n, call.beam_wavelength,
          call.aperture_f_number, _p(r1), _p(r2), call.ray_cone_pitch_ratio, _p(pos), _p(direction), _p(rad))
        return pos, direction, rad

    def pixel_taps(self, camera: dict, x, y):
        """Pixel indices ii/jj [n][4], area weights [n][4] and inside flags of sensor hits (x, y)
        (intersect_sensor, parallel_ray_tracing.cu:1803-1880)."""
        from photon_amd.ray_tracing import camera_from_dict
        cam = camera_from_dict(camera)
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.float32)
        n = x.size
        ii, jj = np.empty((n, 4), np.int32), np.empty((n, 4), np.int32)
        w, inside = np.empty((n, 4), np.float64), np.empty(n, np.int32)
        f = self.lib.oracle_pixel_taps
        f.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 7
        f(n, _p(x), _p(y), ctypes.addressof(cam), _p(ii), _p(jj), _p(w), _p(inside))
        return ii, jj, w, inside.astype(bool)

    # ---- scene generation -----------------------------------------------------------------
    def sources_bos(self, dot_xy, template_xy, z, radiance):
        d = np.ascontiguousarray(dot_xy, dtype=np.float64).reshape(-1, 2)
        t = np.ascontiguousarray(template_xy, dtype=np.float64).reshape(-1, 2)
        dx, dy, tx, ty = (np.ascontiguousarray(a) for a in (d[:, 0], d[:, 1], t[:, 0], t[:, 1]))
        n = d.shape[0] * t.shape[0]
        out = dict(x=np.empty(n, np.float32), y=np.empty(n, np.float32), z=np.empty(n, np.float32),
                   radiance=np.empty(n, np.float64), diameter_index=np.empty(n, np.int32))
        self.lib.oracle_sources_bos(_p(dx), _p(dy), d.shape[0], _p(tx), _p(ty), t.shape[0], ctypes.c_double(z),
                                    ctypes.c_double(radiance), _p(out["x"]), _p(out["y"]), _p(out["z"]),
                                    _p(out["radiance"]), _p(out["diameter_index"]))
        return out

    def sources_piv(self, seed, n, box_min, box_max, z_object, beam_fwhm, irradiance_constant, diameter_cdf=None):
        lo = np.ascontiguousarray(box_min, dtype=np.float64)
        hi = np.ascontiguousarray(box_max, dtype=np.float64)
        cdf = None if diameter_cdf is None else np.ascontiguousarray(diameter_cdf, dtype=np.float64)
        out = dict(x=np.empty(n, np.float32), y=np.empty(n, np.float32), z=np.empty(n, np.float32),
                   radiance=np.empty(n, np.float64), diameter_index=np.empty(n, np.int32))
        self.lib.oracle_sources_piv(ctypes.c_uint64(int(seed)), ctypes.c_longlong(int(n)), _p(lo), _p(hi),
                                    ctypes.c_double(z_object), ctypes.c_double(beam_fwhm),
                                    ctypes.c_double(irradiance_constant), _p(cdf) if cdf is not None else None,
                                    0 if cdf is None else int(cdf.size), _p(out["x"]), _p(out["y"]), _p(out["z"]),
                                    _p(out["radiance"]), _p(out["diameter_index"]))
        return out

    def volume_load_nrrd(self, path, interpolation=1, tex_frac_bits=8):
        h = self.lib.oracle_volume_load_nrrd(path.encode(), interpolation, tex_frac_bits)
        assert h, f"oracle could not read {path}"
        return OracleVolume(self, h)

    # ---- optics ---------------------------------------------------------------------------
    def ray_sphere_intersection(self, center, R, direction, pos, surface: str):
        d = np.ascontiguousarray(direction, np.float32)
        p = np.ascontiguousarray(pos, np.float32)
        c = np.ascontiguousarray(center, np.float32)
        out = np.empty_like(p)
        self.lib.oracle_ray_sphere_intersection(p.shape[0], _p(c), float(R), _p(d), _p(p), surface.encode(), _p(out))
        return out

    def axis_distance(self, pts, center, plane):
        pts = np.ascontiguousarray(pts, np.float32)
        c = np.ascontiguousarray(center, np.float32)
        pl = np.ascontiguousarray(plane, np.float32)
        out = np.empty(pts.shape[0], np.float32)
        self.lib.oracle_axis_distance(pts.shape[0], _p(pts), _p(c), _p(pl), _p(out))
        return out

    def single_element(self, element: dict, center, plane, pos, direction, wavelength, radiance):
        e = element_from_dict(element)
        p = np.array(pos, np.float32, order="C")
        d = np.array(direction, np.float32, order="C")
        r = np.array(radiance, np.float64, order="C")
        c = np.ascontiguousarray(center, np.float32)
        pl = np.ascontiguousarray(plane, np.float32)
        self.lib.oracle_single_element(p.shape[0], ctypes.byref(e), _p(c), _p(pl), _p(p), _p(d), float(wavelength), _p(r))
        return p, d, r

    def optical_system(self, elements, centers, planes, sys_index, pos, direction, wavelength, radiance, train_mode=1):
        """The whole element train on n rays (propagate_rays_through_optical_system, parallel_ray_tracing.cu:1274-1381;
        train_mode 1 = the working train after perform_ray_tracing_03.py:1419-1485)."""
        from photon_amd.ray_tracing import element_data_struct
        n_el = len(elements)
        arr = (element_data_struct * n_el)()
        for k, e in enumerate(elements):
            element_from_dict(e, arr[k])
        p = np.array(pos, np.float32, order="C")
        d = np.array(direction, np.float32, order="C")
        r = np.array(radiance, np.float64, order="C")
        c = np.ascontiguousarray(centers, np.float32).reshape(n_el, 3)
        pl = np.ascontiguousarray(planes, np.float32).reshape(n_el, 4)
        si = np.ascontiguousarray(sys_index, np.int32).reshape(n_el)
        f = self.lib.oracle_optical_system
        f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                      ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]
        f(p.shape[0], ctypes.addressof(arr), _p(c), _p(pl), _p(si), n_el, int(train_mode), _p(p), _p(d), float(wavelength), _p(r))
        return p, d, r


class OracleVolume:
    def __init__(self, o: Oracle, handle):
        self._o, self.handle = o, ctypes.c_void_p(handle)

    def info(self):
        i = photon_volume_info_t()
        self._o.lib.oracle_volume_info(self.handle, ctypes.byref(i))
        return i

    def download(self, coefficients=False):
        i = self.info()
        out = np.empty((i.nz, i.ny, i.nx, 4), np.float32)
        self._o.lib.oracle_volume_download(self.handle, int(coefficients), _p(out))
        return out

    def sample(self, coords):
        c = np.ascontiguousarray(coords, np.float32).reshape(-1, 3)
        out = np.empty((c.shape[0], 4), np.float32)
        self._o.lib.oracle_volume_sample(self.handle, c.shape[0], _p(c), _p(out))
        return out

    def trace_rays(self, pos, direction, algorithm=2):
        p = np.array(pos, np.float32, order="C").reshape(-1, 3)
        d = np.array(direction, np.float32, order="C").reshape(-1, 3)
        steps = np.zeros(p.shape[0], np.int32)
        self._o.lib.oracle_trace_volume_rays(self.handle, int(algorithm), p.shape[0], _p(p), _p(d), _p(steps))
        return p, d, steps

    def free(self):
        if self.handle:
            self._o.lib.oracle_volume_free(self.handle)
            self.handle = None
