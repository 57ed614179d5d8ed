"""C-ABI checks that need no GPU: struct layouts against what the reference's marshalling code
produced (tests/golden/abi_*.{json,npz}), symbol export, header/library agreement."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_fixture_call
from photon_amd import ray_tracing as rt
from photon_amd.library import DECLARED_SYMBOLS


def test_struct_sizes_match_reference_marshalling():
    for case in ("piv", "bos_im1", "bos_im2"):
        with open(os.path.join(GOLDEN, f"abi_{case}.json")) as f:
            sz = json.load(f)["sizeof"]
        assert ctypes.sizeof(rt.scattering_data_struct) == sz["scattering"] == 72
        assert ctypes.sizeof(rt.lightfield_source_struct) == sz["source"] == 64
        assert ctypes.sizeof(rt.camera_design_struct) == sz["camera"] == 112
        assert ctypes.sizeof(rt.element_data_struct) == sz["element"] == 120


def test_field_offsets():
    # SURVEY.md section 8a row a22 (verified there with g++ and ctypes)
    e, c, s, m = rt.element_data_struct, rt.camera_design_struct, rt.lightfield_source_struct, rt.scattering_data_struct
    assert (e.element_type.offset, e.rotation_angles.offset, e.element_properties.offset) == (80, 88, 56)
    assert (c.implement_diffraction.offset, c.rotation_matrix.offset, c.inverse_rotation_matrix.offset) == (36, 40, 76)
    assert (s.num_particles.offset, s.z_offset.offset, s.object_distance.offset) == (48, 52, 56)
    assert (m.scattering_angle.offset, m.scattering_irradiance.offset, m.num_angles.offset) == (48, 56, 64)
    assert ctypes.sizeof(rt.element_geometry_struct) == 32 and ctypes.sizeof(rt.element_properties_struct) == 24


@pytest.mark.parametrize("case", ["piv", "bos_im1", "bos_im2"])
def test_packed_structs_are_byte_identical_to_the_reference(case):
    """Our mirror of prepare_data_for_cytpes_call must lay the camera / element structs out exactly
    as the reference did (raw bytes captured at its ctypes call)."""
    call = load_fixture_call(case)
    a = np.load(os.path.join(GOLDEN, f"abi_{case}.npz"))
    sd, ls, elems, centers, planes, sysidx, cam = call.pack()

    def same(ours: bytes, ref: np.ndarray, holes):
        ours = np.frombuffer(ours, np.uint8).copy()
        ref = ref.copy()
        for lo, hi in holes:                     # padding bytes are unspecified
            ours[lo:hi] = 0
            ref[lo:hi] = 0
        return np.array_equal(ours, ref)

    assert same(bytes(cam), a["raw_camera"], [(37, 40)])
    assert same(bytes(elems[0]), a["raw_element0"], [(21, 24), (29, 32), (52, 56), (81, 84), (116, 120)])
    assert np.array_equal(centers, a["element_center"]) and np.array_equal(planes, a["element_plane_parameters"])
    assert np.array_equal(sysidx, a["element_system_index"].reshape(-1))
    assert ls.num_particles == a["src_x"].size and ls.source_point_number == 10000


def _header_functions():
    with open(os.path.join(ROOT, "include", "parallel_ray_tracing.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b((?:start_ray_tracing|photon_[a-z0-9_]+))\s*\(", text))


def test_header_and_python_binding_declare_the_same_symbols():
    assert _header_functions() == set(DECLARED_SYMBOLS)


def test_library_loads_and_exports_every_declared_symbol():
    """No compute call: just dlopen + dlsym (works without a GPU)."""
    from photon_amd import build
    path = build.build_library()
    lib = ctypes.CDLL(path)
    for name in DECLARED_SYMBOLS:
        assert hasattr(lib, name), name
    lib.photon_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.photon_version()


def test_product_does_not_reference_the_oracle():
    """The product path must not import / link / call anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "photon_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                with open(os.path.join(dirpath, fn), errors="replace") as f:
                    text = f.read()
                assert "oracle_lib" not in text and "libphoton_oracle" not in text and "oracle/" not in text, fn
