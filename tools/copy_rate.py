#!/usr/bin/env python3
"""Device-to-device float4 copy rate at three sizes (the "achievable HBM peak" yardstick of bench.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (first: one HIP runtime per process)
from photon_amd.library import PhotonLibrary  # noqa: E402

lib = PhotonLibrary()
for nb in (1 << 28, 1 << 30, 1 << 31):
    print(nb >> 20, "MiB", round(lib.measure_copy_gbs(nb, 5), 1), "GB/s")
