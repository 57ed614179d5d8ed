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
    """The HIP product library.  GPU tests fail loudly (not skip) if it cannot be loaded."""
    from photon_amd.library import PhotonLibrary
    if not _gpu_available():
        pytest.skip("no GPU in this environment")
    return PhotonLibrary()


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
