#!/usr/bin/env python3
"""Soak of the multi-device path on ONE GPU: many start_ray_tracing calls with PHOTON_DEVICES=0 x K, K and the job varied,
every image held against the single-device image of the same job (f64 accumulation: <= 1e-7 relative L2).  What it is for:
ordering mistakes between the null stream and the workers' non-blocking streams show up once in tens of calls, only while
several shards keep the chip full (round 5: a scene's work queues zeroed by a host-asynchronous hipMemset, 3 of 60 C4 calls).

    python tools/soak_shards.py [--calls 200] [--c4-calls 40]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (first: one HIP runtime per process)
from photon_amd import scenes  # noqa: E402
from photon_amd.library import PhotonLibrary  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=200, help="C3 calls (1e7 rays, 256^3)")
    ap.add_argument("--c4-calls", type=int, default=40, help="C4 calls (1e8 rays, 512^3)")
    args = ap.parse_args()
    lib = PhotonLibrary()
    work = os.path.join(tempfile.gettempdir(), "photon_soak")
    os.makedirs(work, exist_ok=True)
    bad = 0
    for name, n_calls, interps in (("C2", args.calls, ("cubic",)), ("C3", args.calls, ("cubic", "linear")), ("C4", args.c4_calls, ("cubic",))):
        call = scenes.config(name, work, volume_n=512) if name == "C4" else scenes.config(name, work)      # C2: no volume (one fused kernel)
        for interp in interps:
            os.environ["PHOTON_INTERP"] = interp
            os.environ.pop("PHOTON_DEVICES", None)
            one = lib.render(call).astype(np.float64)
            norm = np.linalg.norm(one)
            t0 = time.perf_counter()
            for i in range(n_calls if interp == "cubic" else n_calls // 2):
                k = (2, 3, 5, 8, 8, 8)[i % 6]
                os.environ["PHOTON_DEVICES"] = ",".join(["0"] * k)
                img = lib.render(call).astype(np.float64)
                rel = float(np.linalg.norm(img - one) / norm)
                if not rel <= 1e-7:
                    bad += 1
                    print(f"{name} {interp} call {i} ({k} shards): rel L2 {rel:.3e}, sum {img.sum():.6g} against {one.sum():.6g}", flush=True)
            print(f"{name} {interp}: {n_calls if interp == 'cubic' else n_calls // 2} calls in {time.perf_counter() - t0:.1f} s, {bad} bad so far", flush=True)
    os.environ.pop("PHOTON_DEVICES", None)
    print("SOAK PASS" if bad == 0 else f"SOAK FAIL: {bad} calls differ")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
