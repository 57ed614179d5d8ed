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
