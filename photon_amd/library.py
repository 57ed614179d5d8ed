"""Load libparallel_ray_tracing.so and bind its C-ABI (include/parallel_ray_tracing.h).

``PhotonLibrary`` is a thin ctypes veneer: every method is one call into the HIP library.
There is no Python or CPU fallback -- if the library is missing or a HIP call fails, you get an
exception.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np

from . import build as _build
from .ray_tracing import (RayTracingCall, bind_start_ray_tracing, camera_design_struct, element_data_struct,
                          lightfield_source_struct, scattering_data_struct)

# every symbol include/parallel_ray_tracing.h declares
DECLARED_SYMBOLS = (
    "start_ray_tracing", "photon_set_device", "photon_device_pci_bus_id", "photon_rand_table", "photon_volume_load_nrrd",
    "photon_volume_from_density", "photon_volume_info", "photon_volume_set_weight_bits", "photon_volume_download", "photon_volume_sample",
    "photon_volume_free", "photon_scene_create", "photon_scene_free", "photon_scene_set_noise", "photon_scene_set_element_train", "photon_scene_set_ray_order", "photon_scene_set_skip_doomed", "photon_scene_live_rays", "photon_scene_live_samples", "photon_scene_live_sources", "photon_sources_missing_sensor", "photon_scene_set_source_base", "photon_march_queue_group", "photon_march_queue_count", "photon_march_queue_chunk", "photon_march_queue_size",
    "photon_scene_set_march_segments", "photon_march_segments_plan", "photon_trim_caches", "photon_trace",
    "photon_scene_stats_begin", "photon_scene_stats_end", "photon_scene_check", "photon_scene_set_march_profile", "photon_scene_march_profile", "photon_scene_march_profile_raw",
    "photon_trace_volume_rays", "photon_trace_volume_rays_queued", "photon_version",
    # section 3: scene generation on the device
    "photon_sources_bos", "photon_sources_piv", "photon_sources_count", "photon_sources_download", "photon_sources_free",
    "photon_scene_create_from_sources", "photon_volume_gaussian", "photon_density_gaussian_write_nrrd",
    # section 4: sensor post-processing on the device
    "photon_postprocess_u16", "photon_measure_copy_gbs", "photon_selftest_normal_range_math", "photon_selftest_morton_order",
)


class photon_volume_info_t(ctypes.Structure):
    _fields_ = [("min_bound", ctypes.c_float * 3), ("max_bound", ctypes.c_float * 3),
                ("nx", ctypes.c_int), ("ny", ctypes.c_int), ("nz", ctypes.c_int),
                ("grid_spacing", ctypes.c_float * 3), ("step_size", ctypes.c_float),
                ("data_min", ctypes.c_float), ("interpolation", ctypes.c_int)]


class photon_trace_stats_t(ctypes.Structure):
    _fields_ = [("rays_launched", ctypes.c_uint64), ("rays_on_sensor", ctypes.c_uint64),
                ("rk_iterations", ctypes.c_uint64), ("volume_samples", ctypes.c_uint64),
                ("sensor_taps", ctypes.c_uint64), ("march_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("rays_marched", ctypes.c_uint64), ("shader_clock_mhz", ctypes.c_float), ("traces", ctypes.c_uint32),
                ("march_wave_ms", ctypes.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class photon_march_profile_t(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("launches", ctypes.c_uint32), ("waves", ctypes.c_uint32),
                ("span_ms", ctypes.c_float), ("start_mean_ms", ctypes.c_float), ("start_max_ms", ctypes.c_float),
                ("end_min_ms", ctypes.c_float), ("end_mean_ms", ctypes.c_float)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "struct_size"}
        d["drain_ms"] = d["span_ms"] - d["end_mean_ms"]
        return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items()}


class PhotonError(RuntimeError):
    pass


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _one_hip_runtime_per_process():
    """PyTorch-ROCm bundles its own libamdhip64; the library's RUNPATH names the system's.  A process that ends up with
    both mapped has two HIP runtimes, and whichever touches the GPU second finds no device (seen both ways round on
    the MI355X boxes).  Whoever shares torch device tensors with the library (this repo's tests, bench.py, smoke())
    therefore imports torch FIRST, explicitly: the library then binds to torch's copy by SONAME.  The library itself
    never imports torch behind the caller's back -- photon's own Python has none and gets the system runtime the library
    was built against.  PHOTON_PRELOAD_TORCH=1 asks for the import here (a caller that cannot order its imports)."""
    import sys
    if "torch" in sys.modules or not os.environ.get("PHOTON_PRELOAD_TORCH"):
        return
    try:
        import torch  # noqa: F401
    except Exception:       # noqa: BLE001  -- a broken torch must not keep the library from loading
        pass


def mapped_hip_runtimes():
    """Paths of every libamdhip64 mapped into this process (more than one = the two-runtime trap above)."""
    found = []
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                path = ln.rsplit(" ", 1)[-1].strip()
                if "libamdhip64" in path and path not in found:
                    found.append(path)
    except OSError:
        pass
    return found


class PhotonLibrary:
    def __init__(self, path: Optional[str] = None, build: bool = True):
        if path is None:
            path = os.environ.get("PHOTON_LIBRARY")
            if not path:                    # the in-tree library: (re)built when missing or older than its sources
                path = _build.LIB_PATH
                if build:
                    try:
                        _build.build_library(verbose=False)
                    except Exception as e:  # no hipcc on this box: a library that travelled with the tree is used as is
                        if not os.path.exists(path):
                            raise PhotonError(f"cannot build {path}: {e}") from e
        if not os.path.exists(path):
            raise PhotonError(f"{path} not found: build it with `python -m photon_amd.build` "
                              "(there is no CPU fallback)")
        self.path = path
        _one_hip_runtime_per_process()
        self.lib = ctypes.CDLL(path)
        self.hip_runtimes = mapped_hip_runtimes()
        if os.environ.get("PHOTON_VERBOSE") or len(self.hip_runtimes) > 1:
            import sys
            print(f"photon: {path} runs on {', '.join(self.hip_runtimes) or 'an unidentified HIP runtime'}"
                  + (" -- TWO HIP runtimes in one process: import torch before the library" if len(self.hip_runtimes) > 1 else ""),
                  file=sys.stderr)
        L = self.lib
        self.start_ray_tracing = bind_start_ray_tracing(L)
        L.photon_version.restype = ctypes.c_char_p
        L.photon_set_device.argtypes = [ctypes.c_int]
        L.photon_rand_table.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.photon_volume_load_nrrd.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.photon_volume_from_density.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                 ctypes.POINTER(ctypes.c_void_p)]
        L.photon_volume_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(photon_volume_info_t)]
        L.photon_volume_set_weight_bits.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.photon_volume_download.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.photon_volume_sample.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.photon_volume_free.argtypes = [ctypes.c_void_p]
        L.photon_volume_free.restype = None
        L.photon_scene_create.argtypes = [
            ctypes.c_float, ctypes.c_float, ctypes.POINTER(scattering_data_struct), ctypes.c_char_p,
            ctypes.POINTER(lightfield_source_struct), ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int,
            ctypes.c_void_p, ctypes.POINTER(element_data_struct), ctypes.c_void_p, ctypes.c_void_p,
            ctypes.POINTER(camera_design_struct), ctypes.c_float, ctypes.POINTER(ctypes.c_void_p)]
        L.photon_scene_free.argtypes = [ctypes.c_void_p]
        L.photon_scene_free.restype = None
        L.photon_scene_set_noise.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_float,
                                             ctypes.c_uint64]
        L.photon_scene_set_element_train.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.photon_scene_set_ray_order.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.photon_scene_set_skip_doomed.argtypes = [ctypes.c_void_p, ctypes.c_int]
        if hasattr(L, "photon_scene_set_source_base"):
            L.photon_scene_set_source_base.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        L.photon_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(photon_trace_stats_t)]
        self.has_stats_window = hasattr(L, "photon_scene_stats_begin")     # absent from libraries built before round 3 (A/B runs)
        if self.has_stats_window:
            L.photon_scene_stats_begin.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            L.photon_scene_stats_end.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(photon_trace_stats_t)]
        if hasattr(L, "photon_scene_set_march_segments"):
            L.photon_scene_set_march_segments.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.has_march_profile = hasattr(L, "photon_scene_march_profile")  # round 4
        if self.has_march_profile:
            L.photon_scene_set_march_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
            L.photon_scene_march_profile.argtypes = [ctypes.c_void_p, ctypes.POINTER(photon_march_profile_t)]
        L.photon_trace_volume_rays.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p]
        L.photon_sources_bos.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_void_p)]
        L.photon_sources_piv.argtypes = [ctypes.c_uint64, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_void_p)]
        L.photon_sources_count.argtypes = [ctypes.c_void_p]
        L.photon_sources_count.restype = ctypes.c_longlong
        L.photon_sources_download.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 5
        L.photon_sources_free.argtypes = [ctypes.c_void_p]
        L.photon_sources_free.restype = None
        L.photon_scene_create_from_sources.argtypes = (L.photon_scene_create.argtypes[:5] + [ctypes.c_void_p] +
                                                       L.photon_scene_create.argtypes[5:])
        L.photon_volume_gaussian.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_double,
                                             ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]

        L.photon_postprocess_u16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_float, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]

    # ---- helpers --------------------------------------------------------------------------
    @staticmethod
    def _check(rc: int, what: str):
        if rc != 0:
            raise PhotonError(f"{what} failed with code {rc} (see stderr)")

    def version(self) -> str:
        return self.lib.photon_version().decode()

    def set_device(self, device: int):
        self._check(self.lib.photon_set_device(int(device)), "photon_set_device")

    def rand_table(self, n: int):
        r1 = np.empty(n, np.float32)
        r2 = np.empty(n, np.float32)
        self._check(self.lib.photon_rand_table(n, _ptr(r1), _ptr(r2)), "photon_rand_table")
        return r1, r2

    def morton_order(self, x, y, first: int = 0, n: Optional[int] = None):
        """photon_selftest_morton_order: the spatial order of sources first .. first + n - 1, as the device computes it."""
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        n = x.size - first if n is None else int(n)
        out = np.empty(n, np.int32)
        f = self.lib.photon_selftest_morton_order
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p]
        self._check(f(_ptr(x), _ptr(y), int(x.size), int(first), n, _ptr(out)), "photon_selftest_morton_order")
        return out

    def sources_missing_sensor(self, call: RayTracingCall, lens_x, lens_y):
        """photon_sources_missing_sensor (host arithmetic, no GPU needed): bool[num_sources], True = no ray of that source
        through any of the given lens samples can reach a pixel; None when the call's geometry is not covered."""
        sd, ls, elems, centers, planes, sysidx, cam = call.pack()
        lx = np.ascontiguousarray(lens_x, np.float32)
        ly = np.ascontiguousarray(lens_y, np.float32)
        x, y, z = (np.ascontiguousarray(a, np.float32) for a in (call.src_x, call.src_y, call.src_z))
        off = np.zeros(x.size, np.uint8)
        f = self.lib.photon_sources_missing_sensor
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p,
                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_longlong, ctypes.c_void_p]
        rc = f(_ptr(lx), _ptr(ly), int(lx.size), float(call.image_distance), float(call.beam_wavelength), len(call.elements),
               ctypes.cast(elems, ctypes.c_void_p), _ptr(centers), _ptr(planes), _ptr(sysidx), ctypes.addressof(cam), _ptr(x), _ptr(y), _ptr(z),
               int(x.size), _ptr(off))
        if rc == 1:
            return None
        self._check(rc, "photon_sources_missing_sensor")
        return off.astype(bool)

    # ---- the reference's entry point ----------------------------------------------------------
    def render(self, call: RayTracingCall, image: Optional[np.ndarray] = None) -> np.ndarray:
        """One start_ray_tracing call (host image in, host image out)."""
        if image is None:
            image = call.new_image()
        call.invoke(self.start_ray_tracing, image)
        return image

    def pci_bus_id(self) -> str:
        """PCI bus id of the current device, lower case as sysfs spells it ('0000:c1:00.0'); '' if unavailable."""
        if not hasattr(self.lib, "photon_device_pci_bus_id"):
            return ""
        buf = ctypes.create_string_buffer(64)
        self.lib.photon_device_pci_bus_id.argtypes = [ctypes.c_char_p, ctypes.c_int]
        return buf.value.decode().lower() if self.lib.photon_device_pci_bus_id(buf, 64) == 0 else ""

    def measure_copy_gbs(self, nbytes: int = 1 << 30, reps: int = 5) -> float:
        """Device-to-device float4 copy rate (read + write), GB/s."""
        out = ctypes.c_double(0.0)
        self.lib.photon_measure_copy_gbs.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        self._check(self.lib.photon_measure_copy_gbs(int(nbytes), int(reps), ctypes.byref(out)), "photon_measure_copy_gbs")
        return out.value

    # ---- sensor post-processing on the device (perform_ray_tracing_03.py:2190-2259) -----------------
    def postprocess_u16(self, d_image_ptr: int, width: int, height: int, d_out_ptr: int, pixel_gain: float,
                        pixel_bit_depth: int, intensity_rescaling: bool = True, image_noise: float = 0.0, noise_seed: int = 0,
                        crop_rows: int = 0, crop_cols: int = 0, stream: int = 0):
        """Device f32 image -> device uint16 picture (raw pointers); returns (rows, cols) of the result."""
        r, c = ctypes.c_int(0), ctypes.c_int(0)
        rc = self.lib.photon_postprocess_u16(ctypes.c_void_p(int(d_image_ptr)), int(width), int(height), float(pixel_gain),
                                             int(pixel_bit_depth), int(bool(intensity_rescaling)), float(image_noise),
                                             int(noise_seed), int(crop_rows), int(crop_cols), ctypes.c_void_p(int(d_out_ptr)),
                                             ctypes.byref(r), ctypes.byref(c), ctypes.c_void_p(int(stream)) if stream else None)
        self._check(rc, "photon_postprocess_u16")
        return r.value, c.value

    # ---- volumes ------------------------------------------------------------------------------
    def volume_load_nrrd(self, path: str, interpolation: int = 1) -> "Volume":
        h = ctypes.c_void_p()
        self._check(self.lib.photon_volume_load_nrrd(path.encode(), int(interpolation), ctypes.byref(h)),
                    "photon_volume_load_nrrd")
        return Volume(self, h)

    def volume_from_density(self, rho: np.ndarray, spacing, origin, interpolation: int = 1) -> "Volume":
        """rho indexed [z, y, x] (x fastest), float32."""
        rho = np.ascontiguousarray(rho, dtype=np.float32)
        nz, ny, nx = rho.shape
        sp = np.ascontiguousarray(spacing, dtype=np.float64)
        og = np.ascontiguousarray(origin, dtype=np.float64)
        h = ctypes.c_void_p()
        self._check(self.lib.photon_volume_from_density(_ptr(rho), nx, ny, nz, _ptr(sp), _ptr(og), int(interpolation),
                                                        ctypes.byref(h)), "photon_volume_from_density")
        return Volume(self, h)

    def volume_gaussian(self, n, spacing, origin, rho0: float, amp: float, centre, sigma: float,
                        interpolation: int = 1) -> "Volume":
        """rho0 + amp exp(-|r - centre|^2 / 2 sigma^2) evaluated on the device (no host array, no file)."""
        nx, ny, nz = (n, n, n) if np.isscalar(n) else n
        sp = np.ascontiguousarray(np.broadcast_to(np.asarray(spacing, np.float64), (3,)))
        og = np.ascontiguousarray(origin, dtype=np.float64)
        c = np.ascontiguousarray(centre, dtype=np.float64)
        h = ctypes.c_void_p()
        self._check(self.lib.photon_volume_gaussian(int(nx), int(ny), int(nz), _ptr(sp), _ptr(og), float(rho0), float(amp),
                                                    _ptr(c), float(sigma), int(interpolation), ctypes.byref(h)),
                    "photon_volume_gaussian")
        return Volume(self, h)

    def density_gaussian_write_nrrd(self, path: str, n, spacing, origin, rho0: float, amp: float, centre, sigma: float) -> str:
        """The field of volume_gaussian as an NRRD file (evaluated on the device)."""
        nx, ny, nz = (n, n, n) if np.isscalar(n) else n
        sp = np.ascontiguousarray(np.broadcast_to(np.asarray(spacing, np.float64), (3,)))
        og = np.ascontiguousarray(origin, dtype=np.float64)
        c = np.ascontiguousarray(centre, dtype=np.float64)
        self.lib.photon_density_gaussian_write_nrrd.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double,
                                                                ctypes.c_double, ctypes.c_void_p, ctypes.c_double]
        self._check(self.lib.photon_density_gaussian_write_nrrd(path.encode(), int(nx), int(ny), int(nz), _ptr(sp), _ptr(og),
                                                                float(rho0), float(amp), _ptr(c), float(sigma)),
                    "photon_density_gaussian_write_nrrd")
        return path

    # ---- sources generated on the device ------------------------------------------------------
    def sources_bos(self, dot_xy, template_xy, z: float, radiance: float) -> "Sources":
        d = np.ascontiguousarray(dot_xy, dtype=np.float64).reshape(-1, 2)
        t = np.ascontiguousarray(template_xy, dtype=np.float64).reshape(-1, 2)
        dx, dy, tx, ty = (np.ascontiguousarray(a) for a in (d[:, 0], d[:, 1], t[:, 0], t[:, 1]))
        h = ctypes.c_void_p()
        self._check(self.lib.photon_sources_bos(_ptr(dx), _ptr(dy), d.shape[0], _ptr(tx), _ptr(ty), t.shape[0], float(z),
                                                float(radiance), ctypes.byref(h)), "photon_sources_bos")
        return Sources(self, h)

    def sources_piv(self, seed: int, n: int, box_min, box_max, z_object: float, beam_fwhm: float,
                    irradiance_constant: float, diameter_cdf=None) -> "Sources":
        lo = np.ascontiguousarray(box_min, dtype=np.float64)
        hi = np.ascontiguousarray(box_max, dtype=np.float64)
        cdf = None if diameter_cdf is None else np.ascontiguousarray(diameter_cdf, dtype=np.float64)
        h = ctypes.c_void_p()
        self._check(self.lib.photon_sources_piv(int(seed), int(n), _ptr(lo), _ptr(hi), float(z_object), float(beam_fwhm),
                                                float(irradiance_constant), _ptr(cdf) if cdf is not None else None,
                                                0 if cdf is None else int(cdf.size), ctypes.byref(h)),
                    "photon_sources_piv")
        return Sources(self, h)

    # ---- scenes -------------------------------------------------------------------------------
    def scene_create_from_sources(self, call: RayTracingCall, sources: "Sources") -> "Scene":
        """Like scene_create, with the light-field sources already in HBM (call's own source arrays unused)."""
        sd, ls, elems, centers, planes, sysidx, cam = call.pack()
        h = ctypes.c_void_p()
        rc = self.lib.photon_scene_create_from_sources(
            ctypes.c_float(call.lens_pitch), ctypes.c_float(call.image_distance), ctypes.byref(sd),
            call.scattering_type.encode(), ctypes.byref(ls), sources.handle, int(call.lightray_number_per_particle),
            ctypes.c_float(call.beam_wavelength), ctypes.c_float(call.aperture_f_number), len(call.elements),
            _ptr(centers), elems, _ptr(planes), _ptr(sysidx), ctypes.byref(cam),
            ctypes.c_float(call.ray_cone_pitch_ratio), ctypes.byref(h))
        self._check(rc, "photon_scene_create_from_sources")
        scene = Scene(self, h, call)
        scene.num_sources = sources.count()
        return scene

    def scene_create(self, call: RayTracingCall) -> "Scene":
        sd, ls, elems, centers, planes, sysidx, cam = call.pack()
        h = ctypes.c_void_p()
        rc = self.lib.photon_scene_create(
            ctypes.c_float(call.lens_pitch), ctypes.c_float(call.image_distance), ctypes.byref(sd),
            call.scattering_type.encode(), ctypes.byref(ls), int(call.lightray_number_per_particle),
            ctypes.c_float(call.beam_wavelength), ctypes.c_float(call.aperture_f_number), len(call.elements),
            _ptr(centers), elems, _ptr(planes), _ptr(sysidx), ctypes.byref(cam),
            ctypes.c_float(call.ray_cone_pitch_ratio), ctypes.byref(h))
        self._check(rc, "photon_scene_create")
        return Scene(self, h, call)


class Volume:
    def __init__(self, lib: PhotonLibrary, handle):
        self._lib, self.handle = lib, handle

    def info(self) -> photon_volume_info_t:
        i = photon_volume_info_t()
        self._lib._check(self._lib.lib.photon_volume_info(self.handle, ctypes.byref(i)), "photon_volume_info")
        return i

    def set_weight_bits(self, bits: int):
        """Trilinear weights: 0 = exact f32, 8 = the texture unit's 8 fractional bits."""
        self._lib._check(self._lib.lib.photon_volume_set_weight_bits(self.handle, int(bits)), "photon_volume_set_weight_bits")

    def download(self, coefficients: bool = False) -> np.ndarray:
        i = self.info()
        out = np.empty((i.nz, i.ny, i.nx, 4), np.float32)
        self._lib._check(self._lib.lib.photon_volume_download(self.handle, int(coefficients), _ptr(out)),
                         "photon_volume_download")
        return out

    def sample(self, coords: np.ndarray) -> np.ndarray:
        c = np.ascontiguousarray(coords, dtype=np.float32).reshape(-1, 3)
        out = np.empty((c.shape[0], 4), np.float32)
        self._lib._check(self._lib.lib.photon_volume_sample(self.handle, c.shape[0], _ptr(c), _ptr(out)),
                         "photon_volume_sample")
        return out

    def trace_rays(self, pos: np.ndarray, direction: np.ndarray, algorithm: int = 2):
        p = np.array(pos, dtype=np.float32, order="C").reshape(-1, 3)
        d = np.array(direction, dtype=np.float32, order="C").reshape(-1, 3)
        steps = np.zeros(p.shape[0], np.int32)
        self._lib._check(self._lib.lib.photon_trace_volume_rays(self.handle, int(algorithm), p.shape[0], _ptr(p),
                                                                _ptr(d), _ptr(steps)), "photon_trace_volume_rays")
        return p, d, steps

    def trace_rays_queued(self, pos: np.ndarray, direction: np.ndarray, algorithm: int = 2, segments: int = -1):
        """The march of trace_rays through the render path's launch (persistent waves, work queues, `segments` pieces)."""
        p = np.array(pos, dtype=np.float32, order="C").reshape(-1, 3)
        d = np.array(direction, dtype=np.float32, order="C").reshape(-1, 3)
        f = self._lib.lib.photon_trace_volume_rays_queued
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        self._lib._check(f(self.handle, int(algorithm), p.shape[0], _ptr(p), _ptr(d), int(segments)), "photon_trace_volume_rays_queued")
        return p, d

    def free(self):
        if self.handle:
            self._lib.lib.photon_volume_free(self.handle)
            self.handle = None


class Sources:
    """Light-field sources generated in HBM (photon_sources_t)."""

    def __init__(self, lib: PhotonLibrary, handle):
        self._lib, self.handle = lib, handle

    def count(self) -> int:
        return int(self._lib.lib.photon_sources_count(self.handle))

    def download(self) -> dict:
        n = self.count()
        out = dict(x=np.empty(n, np.float32), y=np.empty(n, np.float32), z=np.empty(n, np.float32),
                   radiance=np.empty(n, np.float64), diameter_index=np.empty(n, np.int32))
        self._lib._check(self._lib.lib.photon_sources_download(self.handle, _ptr(out["x"]), _ptr(out["y"]), _ptr(out["z"]),
                                                               _ptr(out["radiance"]), _ptr(out["diameter_index"])),
                         "photon_sources_download")
        return out

    def free(self):
        if self.handle:
            self._lib.lib.photon_sources_free(self.handle)
            self.handle = None


class Scene:
    def __init__(self, lib: PhotonLibrary, handle, call: RayTracingCall):
        self._lib, self.handle, self.call = lib, handle, call
        self.num_sources = call.num_sources

    def trace(self, d_image_ptr: int, volume: Optional[Volume] = None, algorithm: int = 0, src_begin: int = 0,
              src_end: Optional[int] = None, stream: int = 0, want_stats: bool = False):
        """Accumulate into a DEVICE image (raw pointer, e.g. torch tensor .data_ptr())."""
        if src_end is None:
            src_end = self.num_sources
        stats = photon_trace_stats_t() if want_stats else None
        rc = self._lib.lib.photon_trace(self.handle, volume.handle if volume is not None else None, int(algorithm),
                                        int(src_begin), int(src_end), ctypes.c_void_p(int(d_image_ptr)),
                                        ctypes.c_void_p(int(stream)) if stream else None,
                                        ctypes.byref(stats) if stats is not None else None)
        self._lib._check(rc, "photon_trace")
        return stats

    @property
    def has_stats_window(self) -> bool:
        return self._lib.has_stats_window

    def stats_begin(self, stream: int = 0):
        """Open a statistics window: traces without want_stats record their events and let the counters run."""
        self._lib._check(self._lib.lib.photon_scene_stats_begin(self.handle, ctypes.c_void_p(int(stream)) if stream else None),
                         "photon_scene_stats_begin")

    def stats_end(self, stream: int = 0):
        """Wait for the stream and return the window's sums (photon_trace_stats_t; .traces = calls covered)."""
        stats = photon_trace_stats_t()
        self._lib._check(self._lib.lib.photon_scene_stats_end(self.handle, ctypes.c_void_p(int(stream)) if stream else None,
                                                              ctypes.byref(stats)), "photon_scene_stats_end")
        return stats

    def check(self, stream: int = 0):
        """Wait for `stream`; raise if a trace of this scene had a hand-off error between march segments (photon_scene_check)."""
        f = self._lib.lib.photon_scene_check
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self._lib._check(f(self.handle, ctypes.c_void_p(int(stream)) if stream else None), "photon_scene_check")

    def set_march_segments(self, segments: int):
        """-1 the library's choice, 1 whole marches, n: cut every march of a large launch into n segments (speed only)."""
        self._lib._check(self._lib.lib.photon_scene_set_march_segments(self.handle, int(segments)), "photon_scene_set_march_segments")

    def set_march_profile(self, on: bool):
        """Record wave entry / first-group / exit times of the march launches (measurement; off by default)."""
        self._lib._check(self._lib.lib.photon_scene_set_march_profile(self.handle, int(bool(on))), "photon_scene_set_march_profile")

    def march_profile(self) -> dict:
        """Means over the march launches since the statistics were last reset (see photon_march_profile_t)."""
        out = photon_march_profile_t()
        out.struct_size = ctypes.sizeof(photon_march_profile_t)
        self._lib._check(self._lib.lib.photon_scene_march_profile(self.handle, ctypes.byref(out)), "photon_scene_march_profile")
        return out.as_dict()

    def set_noise(self, add_pos_noise=False, pos_noise_std=0.0, add_ngrad_noise=False, ngrad_noise_std=0.0, seed=0):
        self._lib._check(self._lib.lib.photon_scene_set_noise(self.handle, int(bool(add_pos_noise)), float(pos_noise_std),
                                                              int(bool(add_ngrad_noise)), float(ngrad_noise_std), int(seed)),
                         "photon_scene_set_noise")

    def set_ray_order(self, mode: int):
        """0 source-major (reference order), 1 lens-major over spatially sorted sources, 2 auto (default)."""
        self._lib._check(self._lib.lib.photon_scene_set_ray_order(self.handle, int(mode)), "photon_scene_set_ray_order")

    def set_skip_doomed(self, on: bool):
        """Drop rays that provably die on the first element's aperture before the march (default on)."""
        self._lib._check(self._lib.lib.photon_scene_set_skip_doomed(self.handle, int(bool(on))), "photon_scene_set_skip_doomed")

    def live_rays(self) -> int:
        """Lens samples per source a volume-free launch keeps (photon_scene_live_rays); rays_per_source = nothing ruled out."""
        f = self._lib.lib.photon_scene_live_rays
        f.argtypes = [ctypes.c_void_p]
        f.restype = ctypes.c_int
        return int(f(self.handle))

    def live_sources(self):
        """The sources a volume-free launch keeps, ascending (photon_scene_live_sources); None = every source."""
        f = self._lib.lib.photon_scene_live_sources
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
        f.restype = ctypes.c_longlong
        n = f(self.handle, None, 0)
        if n == -1:
            return None
        if n < 0:
            raise PhotonError(f"photon_scene_live_sources returned {n}")
        out = np.empty(int(n), np.int32)
        got = f(self.handle, out.ctypes.data_as(ctypes.c_void_p), int(n))
        if got != n:
            raise PhotonError(f"photon_scene_live_sources returned {got}, expected {n}")
        return out

    def live_samples(self):
        """The lens samples a volume-free launch keeps, ascending (photon_scene_live_samples)."""
        import numpy as np
        n = self.live_rays()
        out = np.zeros(max(n, 1), np.int32)
        f = self._lib.lib.photon_scene_live_samples
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        f.restype = ctypes.c_int
        got = int(f(self.handle, out.ctypes.data, int(out.size)))
        if got != n:
            raise PhotonError(f"photon_scene_live_samples returned {got}, expected {n}")
        return out[:n]

    def set_source_base(self, first_source: int):
        """This scene holds the slice of a job's sources that starts at `first_source` (noise ids stay job-wide)."""
        self._lib._check(self._lib.lib.photon_scene_set_source_base(self.handle, int(first_source)), "photon_scene_set_source_base")

    def set_element_train(self, mode: int):
        """0 = the reference's element walk (element 0 only), 1 = the working multi-element train."""
        self._lib._check(self._lib.lib.photon_scene_set_element_train(self.handle, int(mode)),
                         "photon_scene_set_element_train")

    def free(self):
        if self.handle:
            self._lib.lib.photon_scene_free(self.handle)
            self.handle = None
