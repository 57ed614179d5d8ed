"""Where one start_ray_tracing call spends its time beside the trace (PHOTON_VERBOSE phase lines), on the C3 job:
    python tools/abi_call_breakdown.py [dots] [interp]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from photon_amd import scenes  # noqa: E402
from photon_amd.library import PhotonLibrary  # noqa: E402

dots = int(sys.argv[1]) if len(sys.argv) > 1 else 200
interp = sys.argv[2] if len(sys.argv) > 2 else "cubic"
os.environ["PHOTON_VERBOSE"] = "1"
os.environ["PHOTON_INTERP"] = interp
lib = PhotonLibrary()
work = os.path.join(tempfile.gettempdir(), "photon_bench")
os.makedirs(work, exist_ok=True)
call = scenes.config("C3", work) if dots == 200 else scenes.bos_scene(n_dots=dots, density_grad_filename=scenes.config("C3", work).density_grad_filename)
for k in range(4):
    t0 = time.perf_counter()
    lib.render(call)
    print(f"call {k}: {(time.perf_counter() - t0) * 1e3:.2f} ms through ctypes", flush=True)
