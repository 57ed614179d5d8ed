"""photon_amd - MI355X-native ray-tracing core for photon (hot path only).

csrc/           hand-written HIP kernels + the C-ABI (libparallel_ray_tracing.so)
ray_tracing.py  host-side mirror of photon's ctypes marshalling layer
library.py      build / load the shared library, bind the photon_* extension entry points
scenes.py       synthetic PIV / BOS workloads of BASELINE.json and the NRRD writer
"""
__version__ = "0.1.0"
