"""The drop-in boundary exercised by the REAL caller: photon's unmodified Python (`run_simulation_02.py` ->
`perform_ray_tracing_03.py`) loads this repo's library from the path it always uses and calls `start_ray_tracing` with
its own marshalling (tests/golden/dropin_check.py).  Needs /root/reference, so it runs only in the build container (the
GPU boxes do not have the reference: there the call path is covered by the captured ABI fixtures).  Without a GPU the
library must report the HIP error and leave the image untouched -- the contract of the header for a failed call."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/python_codes"), reason="the reference tree exists only in the build container")
@pytest.mark.skipif(os.environ.get("PHOTON_RUN_REFERENCE", "1") == "0",
                    reason="PHOTON_RUN_REFERENCE=0: do not execute the reference tree's Python in this run")
def test_reference_python_loads_and_calls_the_library():
    """Executes the (public, untrusted) reference tree's own Python in a CHILD process -- on by default where the tree
    exists, because that is the only place the real caller can be exercised; PHOTON_RUN_REFERENCE=0 keeps a test run
    hermetic.  Skips, rather than fails, where the library can neither be found nor built."""
    from photon_amd.library import PhotonError, PhotonLibrary
    try:
        PhotonLibrary()                                 # builds the library if it is missing or stale
    except PhotonError as e:
        pytest.skip(f"no library and no way to build one here: {e}")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "dropin_check.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=600)
    out, err = r.stdout.decode("utf-8", "replace"), r.stderr.decode("utf-8", "replace")
    assert r.returncode == 0, err[-2000:]
    line = [ln for ln in out.splitlines() if ln.startswith("DROPIN ")]
    assert len(line) == 1
    d = json.loads(line[0][len("DROPIN "):])
    lib = os.path.realpath(os.path.join(ROOT, "photon_amd", "libparallel_ray_tracing.so"))
    # the sample BOS case renders two images: two loads through photon's own relative path, both landing on our file
    assert len(d["cdll_calls"]) == 2 and d["images_returned"] == 2
    for c in d["cdll_calls"]:
        assert c["path"].endswith("cuda_codes/Debug/libparallel_ray_tracing.so") and c["resolves_to"] == lib and c["has_symbol"]
    if not os.path.exists("/dev/kfd"):
        # no GPU here: each call says so and hands the caller's image back untouched; photon's driver carries on
        assert err.count("image left untouched") == 2 and "photon: HIP error" in err
        assert d["raw_image_sums"] == [0.0, 0.0]
    else:
        assert all(s > 0 for s in d["raw_image_sums"])
